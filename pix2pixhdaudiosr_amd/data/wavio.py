"""RIFF/WAVE reader for the feeder (stands in for torchaudio.info / torchaudio.load of data/audio_dataset.py:31-38,99):
header parse + a seek to the requested frame window, so a random training segment costs one read of segment_length
frames, not the file.  PCM 8/16/24/32-bit and IEEE float, WAVE_FORMAT_EXTENSIBLE included; samples are scaled the way
`torchaudio.load(normalize=True)` documents (signed PCM / 2^(bits-1)).  Host I/O only -- no arithmetic of the hot path."""
import struct
from collections import namedtuple

import numpy as np
import torch

WavInfo = namedtuple("WavInfo", "sample_rate num_frames num_channels bits_per_sample format_tag data_offset block_align")


def info(path):
    with open(path, "rb") as f:
        head = f.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                raise ValueError(f"{path}: no data chunk")
            cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            if cid == b"fmt ":
                fmt = f.read(size)
                if size & 1:
                    f.seek(1, 1)
            elif cid == b"data":
                if fmt is None:
                    raise ValueError(f"{path}: data chunk before fmt chunk")
                tag, ch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
                if tag == 0xFFFE and len(fmt) >= 26:
                    tag = struct.unpack("<H", fmt[24:26])[0]
                if tag not in (1, 3) or bits not in (8, 16, 24, 32, 64) or ch < 1 or align != ch * bits // 8:
                    raise ValueError(f"{path}: unsupported WAVE format (tag {tag}, {bits} bit, {ch} ch)")
                return WavInfo(rate, size // align, ch, bits, tag, f.tell(), align)
            else:
                f.seek(size + (size & 1), 1)


def load(path, frame_offset=0, num_frames=-1):
    """-> (float32 tensor [channels, frames], sample_rate), like torchaudio.load."""
    meta = info(path)
    start = min(max(int(frame_offset), 0), meta.num_frames)
    stop = meta.num_frames if num_frames < 0 else min(meta.num_frames, start + int(num_frames))
    with open(path, "rb") as f:
        f.seek(meta.data_offset + start * meta.block_align)
        raw = f.read((stop - start) * meta.block_align)
    bits = meta.bits_per_sample
    if meta.format_tag == 3:
        a = np.frombuffer(raw, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif bits == 8:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) * (1.0 / 128.0)
    elif bits == 16:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) * (1.0 / 32768.0)
    elif bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)) << 8                 # sign bit into bit 31
        a = (v >> 8).astype(np.float32) * (1.0 / 8388608.0)
    else:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) * (1.0 / 2147483648.0)
    return torch.from_numpy(np.ascontiguousarray(a.reshape(-1, meta.num_channels).T)), meta.sample_rate


def save(path, waveform, sample_rate):
    """PCM16 writer (torchaudio.save's default for float input is float32; the reference's outputs are listened to,
    not re-read, generate_audio.py:65): waveform [channels, frames] or [frames] in [-1, 1)."""
    w = torch.as_tensor(waveform).detach().cpu().float()
    if w.dim() == 1:
        w = w.unsqueeze(0)
    pcm = (w.clamp(-1.0, 32767.0 / 32768.0) * 32768.0).round().to(torch.int16).T.contiguous().numpy().tobytes()
    ch = w.shape[0]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, int(sample_rate), int(sample_rate) * ch * 2, ch * 2, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)

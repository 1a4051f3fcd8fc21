"""Oracle: input feeder -- wav decode, segmenting and the HR->LR->HR sinc resampler in numpy fp64.  TEST INFRASTRUCTURE ONLY.

Restates (reference paths):
  AudioDataset.readaudio / __getitem__   data/audio_dataset.py:30-62    random segment of a file, resample to the HR rate and
                                                                         through the LR rate and back, seg_pad both
  AudioDataset.seg_pad_audio             data/audio_dataset.py:81-88    crop to segment_length (first channel) or right-pad
  AudioTestDataset.__init__              data/audio_dataset.py:99-114   whole file -> (LR ->) HR rate -> segments
  AudioTestDataset.seg_pad_audio         data/audio_dataset.py:124-135  right-pad to a multiple of segment_length, unfold

Third-party arithmetic absent from /root/reference: `torchaudio.functional.resample` and `torchaudio.load` (torchaudio is
unpinned in the reference and not installed here).  PARITY UNPINNED for the resampler: this file states
the build's own definition -- band-limited interpolation with a Hann-windowed sinc, lowpass_filter_width 6, rolloff 0.99
(the documented torchaudio defaults): with o = orig/gcd, n = new/gcd, base = min(o, n) * rolloff,
width = ceil(lpw * o / base), for phase p < n and tap k < 2*width + o:
    t = clamp((-p/n + (k - width)/o) * base, -lpw, lpw);  h[p][k] = sinc(t) * cos^2(pi t / (2 lpw)) * base / o
    y[j*n + p] = sum_k h[p][k] * x[j*o + k - width]      (x zero outside [0, T)),   len(y) = ceil(n*T/o).
`seg_pad` is pinned by golden vectors from the reference methods themselves; wav decode by PCM16 scaling 1/32768.
"""
import math
import struct

import numpy as np


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = int(math.ceil(lowpass_filter_width * o / base))
    idx = np.arange(-width, width + o, dtype=np.float64)[None, :] / o
    t = (np.arange(0, -n, -1, dtype=np.float64)[:, None] / n + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    tp = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(tp == 0, 1.0, np.sin(tp) / tp)
    return k * window * (base / o), width, o, n


def resample(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    x = np.asarray(x, np.float64)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    h, width, o, n = resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    T = x2.shape[-1]
    xp = np.pad(x2, [(0, 0), (width, width + o)])
    J = (xp.shape[-1] - h.shape[1]) // o + 1
    idx = np.arange(J)[:, None] * o + np.arange(h.shape[1])[None, :]
    y = np.einsum("bjk,pk->bjp", xp[:, idx], h).reshape(x2.shape[0], -1)
    return y[:, : int(math.ceil(n * T / o))].reshape(lead + (-1,))


def seg_pad_train(waveform, segment_length):
    """data/audio_dataset.py:81-88.  waveform [C, T]: long input -> 1-D first channel crop; short -> [C, segment] padded."""
    waveform = np.asarray(waveform)
    if waveform.shape[1] >= segment_length:
        return waveform[0][:segment_length]
    return np.pad(waveform, [(0, 0), (0, segment_length - waveform.shape[1])])


def seg_pad_test(audio, segment_length):
    """data/audio_dataset.py:124-135.  audio [1, T] or [T] -> [num_segments, segment_length]."""
    audio = np.asarray(audio)
    if audio.ndim == 2 and audio.shape[0] == 1:
        audio = audio[0]
    length = len(audio)
    if length >= segment_length:
        ns = int(np.ceil(length / segment_length))
        audio = np.pad(audio, (0, segment_length * ns - length))
        return audio.reshape(ns, segment_length)
    return np.pad(audio, (0, segment_length - length))[None]


def read_wav(path, frame_offset=0, num_frames=-1):
    """RIFF/WAVE decode to float32 [channels, frames] in [-1, 1) the way torchaudio.load(normalize=True) scales PCM:
    8-bit unsigned (x-128)/128, 16/24/32-bit signed / 2^(bits-1), IEEE float passthrough."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE", "not a RIFF/WAVE file"
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    tag, ch, rate, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
    if tag == 0xFFFE:
        tag = struct.unpack("<H", fmt[24:26])[0]
    n = len(pcm) // align
    if tag == 3:
        a = np.frombuffer(pcm[: n * align], dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    elif bits == 8:
        a = (np.frombuffer(pcm[: n * align], dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif bits == 16:
        a = np.frombuffer(pcm[: n * align], dtype="<i2").astype(np.float32) / 32768.0
    elif bits == 24:
        b = np.frombuffer(pcm[: n * align], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        a = v.astype(np.float32) / float(1 << 23)
    else:
        a = np.frombuffer(pcm[: n * align], dtype="<i4").astype(np.float32) / float(1 << 31)
    a = a.reshape(n, ch).T
    end = a.shape[1] if num_frames < 0 else min(a.shape[1], frame_offset + num_frames)
    return np.ascontiguousarray(a[:, frame_offset:end]), rate

"""Feeder: oracle pinned to the reference's seg_pad_audio outputs; host-side pieces of the product
(wav decode, segmenting, the C-ABI phase-table fill) against the oracle.  No GPU."""
import os
import struct
import types
import wave

import numpy as np
import pytest
import torch

from oracle import feeder as OF

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "feeder.npz"))
TRAIN = ["train_long", "train_exact", "train_short", "train_stereo_long"]
TEST = ["test_multi", "test_exact", "test_short", "test_1d"]


@pytest.mark.parametrize("tag", TRAIN + TEST)
def test_seg_pad_oracle_and_product_match_reference(tag):
    from pix2pixhdaudiosr_amd.data.audio_dataset import AudioDataset, AudioTestDataset
    w, ref, seg = G[f"seg_{tag}_in"], G[f"seg_{tag}_out"], int(G[f"seg_{tag}_len"])
    o = types.SimpleNamespace(segment_length=seg)
    if tag in TRAIN:
        got_o = OF.seg_pad_train(w, seg)
        got_p = AudioDataset.seg_pad_audio(o, torch.from_numpy(w)).numpy()
    else:
        got_o = OF.seg_pad_test(w, seg)
        got_p = AudioTestDataset.seg_pad_audio(o, torch.from_numpy(w)).numpy()
    assert got_o.shape == ref.shape and np.array_equal(got_o, ref)
    assert got_p.shape == ref.shape and np.array_equal(got_p, ref)


def test_resample_oracle_properties():
    t = np.arange(48000) / 48000.0
    x = np.sin(2 * np.pi * 1000 * t) + 0.5 * np.sin(2 * np.pi * 2000 * t)
    lo = OF.resample(x, 48000, 8000)
    assert lo.shape == (8000,)
    t8 = np.arange(8000) / 8000.0
    assert np.abs(lo - (np.sin(2 * np.pi * 1000 * t8) + 0.5 * np.sin(2 * np.pi * 2000 * t8)))[200:-200].max() < 1e-2
    up = OF.resample(lo, 8000, 48000)
    assert up.shape == (48000,) and np.abs(up - x)[600:-600].max() < 1.5e-2
    hi = np.sin(2 * np.pi * 9000 * t)                                   # above the LR Nyquist: removed by the round trip
    assert np.abs(OF.resample(OF.resample(hi, 48000, 8000), 8000, 48000))[600:-600].max() < 2e-3
    assert OF.resample(x[:1001], 44100, 48000).shape == (int(np.ceil(160 * 1001 / 147)),)
    assert np.array_equal(OF.resample(x, 48000, 48000), x)
    k, width, o, n = OF.resample_kernel(48000, 8000)
    assert (width, o, n, k.shape) == (37, 6, 1, (1, 80))
    k, width, o, n = OF.resample_kernel(8000, 48000)
    assert (width, o, n, k.shape) == (7, 1, 6, (6, 15))


@pytest.mark.parametrize("rates", [(48000, 8000), (8000, 48000), (44100, 48000), (48000, 16000), (22050, 48000)])
def test_phase_table_fill_matches_oracle(rates):
    import ctypes as C
    from pix2pixhdaudiosr_amd import _lib
    L = _lib.lib()
    k, width, o, n = OF.resample_kernel(*rates)
    go, gn, gw, gk = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert L.p2phd_resample_geometry(rates[0], rates[1], 6, 0.99, C.byref(go), C.byref(gn), C.byref(gw), C.byref(gk)) == 0
    assert (go.value, gn.value, gw.value, gk.value) == (o, n, width, k.shape[1])
    buf = torch.empty(L.p2phd_resample_kernel_floats(rates[0], rates[1], 6, 0.99), dtype=torch.float32)
    assert buf.numel() == k.size
    assert L.p2phd_resample_kernel_fill(rates[0], rates[1], 6, 0.99, _lib.ptr(buf)) == 0
    assert np.abs(buf.numpy().reshape(k.shape) - k).max() <= 1e-7
    assert L.p2phd_resample_out_len(1001, rates[0], rates[1]) == int(np.ceil(n * 1001 / o))
    assert L.p2phd_resample_geometry(0, 8000, 6, 0.99, None, None, None, None) != 0
    assert b"positive" in L.p2phd_last_error()


def _write_wav(path, data, rate, bits, tag=1, extensible=False):
    ch = data.shape[0]
    inter = data.T.reshape(-1)
    if tag == 3:
        raw = inter.astype("<f4").tobytes()
    elif bits == 8:
        raw = inter.astype(np.uint8).tobytes()
    elif bits == 16:
        raw = inter.astype("<i2").tobytes()
    elif bits == 24:
        v = inter.astype(np.int32)
        raw = b"".join(struct.pack("<i", int(s))[:3] for s in v)
    else:
        raw = inter.astype("<i4").tobytes()
    align = ch * bits // 8
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, ch, rate, rate * align, align, bits, 22, bits, 0, tag, b"\x00" * 14)
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * align, align, bits)
    junk = b"LIST" + struct.pack("<I", 5) + b"abcde\x00"                    # odd-sized chunk + pad byte before fmt
    body = b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", len(raw)) + raw
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


@pytest.mark.parametrize("bits,tag,ext", [(16, 1, False), (24, 1, False), (32, 1, False), (8, 1, False), (32, 3, False), (16, 1, True)])
def test_wav_decode(tmp_path, bits, tag, ext):
    from pix2pixhdaudiosr_amd.data import wavio
    rng = np.random.default_rng(5)
    if tag == 3:
        data = rng.uniform(-1, 1, size=(2, 333)).astype(np.float32)
    elif bits == 8:
        data = rng.integers(0, 256, size=(2, 333))
    else:
        data = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=(2, 333))
    p = str(tmp_path / "a.wav")
    _write_wav(p, data, 44100, bits, tag, ext)
    meta = wavio.info(p)
    assert (meta.sample_rate, meta.num_frames, meta.num_channels, meta.bits_per_sample) == (44100, 333, 2, bits)
    ref, rate = OF.read_wav(p)
    got, rate2 = wavio.load(p)
    assert rate == rate2 == 44100 and got.dtype == torch.float32 and tuple(got.shape) == (2, 333)
    assert np.array_equal(got.numpy(), ref)
    expect = data.astype(np.float64) if tag == 3 else ((data - 128) / 128.0 if bits == 8 else data / float(1 << (bits - 1)))
    assert np.abs(got.numpy() - expect).max() <= 1e-7
    part, _ = wavio.load(p, frame_offset=100, num_frames=50)
    assert np.array_equal(part.numpy(), ref[:, 100:150])
    tail, _ = wavio.load(p, frame_offset=300, num_frames=100)
    assert tuple(tail.shape) == (2, 33)


def test_wav_roundtrip_reference_clip_and_errors(tmp_path):
    from pix2pixhdaudiosr_amd.data import wavio
    pcm = G["test_wav_excerpt_i16"]
    p = str(tmp_path / "clip.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(48000); w.writeframes(pcm.tobytes())
    x, rate = wavio.load(p)
    assert rate == 48000 and np.array_equal(x.numpy()[0], pcm.astype(np.float32) / 32768.0)
    q = str(tmp_path / "back.wav")
    wavio.save(q, x, 48000)
    with wave.open(q) as w:
        assert np.array_equal(np.frombuffer(w.readframes(w.getnframes()), dtype="<i2"), pcm)
    bad = str(tmp_path / "bad.wav")
    open(bad, "wb").write(b"not a wave file at all")
    with pytest.raises(ValueError):
        wavio.info(bad)


def test_audio_dataset_items_are_io_only(tmp_path):
    """Worker-side __getitem__: a random window of segment_length frames (or the whole short file), zero padded."""
    from pix2pixhdaudiosr_amd.data import wavio
    from pix2pixhdaudiosr_amd.data.audio_dataset import AudioDataset
    pcm = G["test_wav_excerpt_i16"].astype(np.float32) / 32768.0
    wavio.save(str(tmp_path / "long.wav"), torch.from_numpy(pcm), 48000)
    wavio.save(str(tmp_path / "short.wav"), torch.from_numpy(pcm[:1000]), 44100)
    opt = types.SimpleNamespace(lr_sampling_rate=8000, hr_sampling_rate=48000, segment_length=4096, n_fft=64, hop_length=32,
                                win_length=64, dataroot=str(tmp_path), center=True, seed=1234)
    ds = AudioDataset(opt)
    assert len(ds) == 2
    items = {os.path.basename(ds[i]['path']): ds[i] for i in range(2)}
    lg, sh = items["long.wav"], items["short.wav"]
    assert lg['rate'] == 48000 and lg['raw_len'] == 4096 and tuple(lg['raw'].shape) == (4096,)
    ref = torch.from_numpy(pcm)
    starts = [s for s in range(len(pcm) - 4096) if ref[s] == lg['raw'][0] and torch.equal(ref[s:s + 4096], lg['raw'])]
    assert len(starts) >= 1                                      # the item is a contiguous window of the file
    assert sh['rate'] == 44100 and sh['raw_len'] == 1000 and torch.equal(sh['raw'][:1000], ref[:1000]) and not sh['raw'][1000:].any()

"""HIP feeder (csrc/resample.hip, data/audio_dataset.py) through the C ABI against the oracle (oracle/feeder.py)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "feeder.npz"))
TOL = 2e-6        # fp32 accumulation of <= 161 taps on |x| <= 1 against the fp64 oracle, absolute


@pytest.mark.parametrize("rates", [(48000, 8000), (8000, 48000), (44100, 48000), (48000, 44100), (16000, 48000), (48000, 12000)])
@pytest.mark.parametrize("shape", [(3, 5003), (1, 37), (2, 1)])
def test_resample_matches_oracle(rates, shape):
    from oracle import feeder as OF
    from pix2pixhdaudiosr_amd.data.resample import resample
    g = torch.Generator().manual_seed(shape[1] + rates[0])
    x = torch.rand(*shape, generator=g) * 2 - 1
    y = resample(x.cuda(), *rates)
    ref = OF.resample(x.numpy(), *rates)
    assert tuple(y.shape) == ref.shape
    assert np.abs(y.cpu().numpy() - ref).max() <= TOL


def test_resample_shapes_identity_and_errors():
    from pix2pixhdaudiosr_amd import _lib
    from pix2pixhdaudiosr_amd.data.resample import resample
    x = torch.randn(2, 3, 100, device="cuda")
    assert resample(x, 48000, 48000) is x
    assert tuple(resample(x, 48000, 8000).shape) == (2, 3, 17)
    assert tuple(resample(x[..., :0], 48000, 8000).shape) == (2, 3, 0)
    with pytest.raises(_lib.P2PHDError):
        resample(x, 48000, 0)
    with pytest.raises((TypeError, ValueError, _lib.P2PHDError)):
        resample(x.cpu(), 48000, 8000)                              # no CPU path


def test_full_size_batch_properties_and_row_parity():
    """BASELINE batch geometry (32 x 130 560 at 48 kHz -> 8 kHz -> 48 kHz): linearity, row independence, one row
    against the oracle."""
    from oracle import feeder as OF
    from pix2pixhdaudiosr_amd.data.resample import resample
    g = torch.Generator().manual_seed(8)
    a = (0.1 * torch.randn(32, 130560, generator=g)).cuda()
    b = (0.1 * torch.randn(32, 130560, generator=g)).cuda()
    rt = lambda x: resample(resample(x, 48000, 8000), 8000, 48000)
    ya, yb, yab = rt(a), rt(b), rt(a + 2 * b)
    assert tuple(ya.shape) == (32, 130560)
    assert (yab - (ya + 2 * yb)).abs().max().item() <= 5e-6
    assert torch.equal(rt(a[5:6]), ya[5:6])
    ref = OF.resample(OF.resample(a[7].cpu().numpy(), 48000, 8000), 8000, 48000)
    assert np.abs(ya[7].cpu().numpy() - ref).max() <= TOL


def test_gpu_feeder_equals_per_item_reference_semantics():
    """A batch of mixed rates / lengths through GpuFeeder == each un-padded item resampled on its own and seg_pad'ed
    (AudioDataset.__getitem__, data/audio_dataset.py:55-61)."""
    from oracle import feeder as OF
    from pix2pixhdaudiosr_amd.data.audio_dataset import GpuFeeder
    seg = 6000
    opt = types.SimpleNamespace(lr_sampling_rate=8000, hr_sampling_rate=48000, segment_length=seg)
    pcm = G["test_wav_excerpt_i16"].astype(np.float32) / 32768.0
    items = [(pcm[:seg], 48000), (pcm[1000:1000 + 4321], 48000), (pcm[3000:3000 + seg], 44100), (pcm[500:500 + 777], 44100),
             (pcm[9000:9000 + seg], 16000)]
    raw = torch.zeros(len(items), seg)
    for i, (x, _) in enumerate(items):
        raw[i, :len(x)] = torch.from_numpy(x)
    batch = {'raw': raw, 'raw_len': torch.tensor([len(x) for x, _ in items]), 'rate': torch.tensor([r for _, r in items]),
             'path': ['p'] * len(items), 'inst': torch.zeros(len(items)), 'feat': torch.zeros(len(items))}
    out = GpuFeeder(opt, 'cuda')(batch)
    assert set(out) == {'image', 'label', 'path', 'inst', 'feat'}
    assert tuple(out['image'].shape) == tuple(out['label'].shape) == (len(items), seg)
    for i, (x, rate) in enumerate(items):
        hr = OF.seg_pad_train(OF.resample(x[None], rate, 48000), seg)
        lr = OF.seg_pad_train(OF.resample(OF.resample(x[None], rate, 8000), 8000, 48000), seg)
        assert np.abs(out['image'][i].cpu().numpy() - hr.reshape(-1)).max() <= TOL, i
        assert np.abs(out['label'][i].cpu().numpy() - lr.reshape(-1)).max() <= TOL, i


def test_datasets_end_to_end(tmp_path):
    from oracle import feeder as OF
    from pix2pixhdaudiosr_amd.data import wavio
    from pix2pixhdaudiosr_amd.data.audio_dataset import AudioTestDataset
    from pix2pixhdaudiosr_amd.data.data_loader import CreateDataLoader
    pcm = G["test_wav_excerpt_i16"].astype(np.float32) / 32768.0
    clip = str(tmp_path / "clip.wav")
    wavio.save(clip, torch.from_numpy(pcm), 48000)
    base = dict(lr_sampling_rate=8000, hr_sampling_rate=48000, segment_length=4064, n_fft=64, hop_length=32, win_length=64,
                center=True, seed=1234, batchSize=2, nThreads=0, max_dataset_size=float("inf"), serial_batches=True)
    # test phase: whole file -> LR round trip -> segments on the GPU
    opt = types.SimpleNamespace(dataroot=clip, phase='test', is_lr_input=False, **base)
    ds = AudioTestDataset(opt)
    lr = OF.resample(OF.resample(pcm[None], 48000, 8000), 8000, 48000)
    segs = OF.seg_pad_test(lr, 4064)
    assert ds.raw_audio.is_cuda and tuple(ds.seg_audio.shape) == segs.shape == (6, 4064) and len(ds) == 6
    assert np.abs(ds.seg_audio.cpu().numpy() - segs).max() <= TOL
    loader = CreateDataLoader(opt)
    batches = list(loader.load_data())
    assert len(batches) == 3 and tuple(batches[0]['label'].shape) == (2, 4064) and batches[0]['label'].is_cuda
    assert np.abs(torch.cat([b['label'] for b in batches]).cpu().numpy() - segs).max() <= TOL
    opt_lr = types.SimpleNamespace(dataroot=clip, phase='test', is_lr_input=True, **base)
    assert torch.equal(AudioTestDataset(opt_lr).lr_audio, ds.raw_audio)       # already at the HR rate: untouched
    # train phase: directory of clips -> fed batches
    d = tmp_path / "train"
    d.mkdir()
    for i in range(5):
        wavio.save(str(d / f"{i}.wav"), torch.from_numpy(pcm[i * 3000: i * 3000 + 9000]), 48000)
    opt_tr = types.SimpleNamespace(dataroot=str(d), phase='train', validation_split=0.2, val_indices=None,
                                   checkpoints_dir=str(tmp_path), name='run', **base)
    tl = CreateDataLoader(opt_tr)
    assert len(tl) == 4 and tl.eval_data_len() == 1 and os.path.isfile(tmp_path / "run" / "validation_indices.pt")
    n = 0
    for b in tl.load_data():
        assert b['image'].is_cuda and b['image'].shape[1] == 4064 and b['image'].shape == b['label'].shape
        hr = b['image'].cpu().numpy()
        for row in hr:                                                # HR rate == file rate: the item is a window of the clip
            assert any(np.array_equal(row, pcm[s:s + 4064]) for s in np.flatnonzero(pcm == row[0]))
        n += hr.shape[0]
    assert n == 4
    assert sum(b['image'].shape[0] for b in tl.eval_data()) == 1

"""Input feeder with the reference's dataset names (data/audio_dataset.py).

MI355X-first split of the work: DataLoader workers only do file I/O (a seek + one read of `segment_length` frames,
`wavio.load`); the arithmetic the reference does per item on the CPU -- three `torchaudio.functional.resample` calls
and `seg_pad_audio` (audio_dataset.py:55-60) -- runs per BATCH on the GPU (`GpuFeeder`, csrc/resample.hip), so the
feeder keeps up with a training step that consumes ~800 segments/s per GPU.  Results per item are identical to
resampling each un-padded file on its own: rows are masked to ceil(new*len/orig) after every stage, which is where
the per-item output would have ended.
"""
import csv
import math
import os

import torch
import torch.nn.functional as F
import torch.utils.data as data

from . import wavio
from .resample import resample


class BaseDataset(data.Dataset):
    def name(self):
        return 'BaseDataset'

    def initialize(self, opt):
        pass


def _masked_resample(x, lengths, orig, new):
    """Batched resample of right-padded rows; each row is cut (zeroed) where its own un-padded output would end."""
    if int(orig) == int(new):
        return x, lengths
    y = resample(x, orig, new)
    g = math.gcd(int(orig), int(new))
    o, n = int(orig) // g, int(new) // g
    out_len = (lengths * n + o - 1) // o
    y = y * (torch.arange(y.shape[-1], device=y.device)[None, :] < out_len[:, None])
    return y, out_len


def _fit(x, segment_length):
    """seg_pad_audio for a batch (audio_dataset.py:81-88): crop or right-pad the last dim to segment_length."""
    T = x.shape[-1]
    if T >= segment_length:
        return x[..., :segment_length].contiguous()
    return F.pad(x, (0, segment_length - T))


class GpuFeeder:
    """raw batch {'raw' [B, segment_length] at `rate`, 'raw_len' [B]} -> {'image': hr, 'label': lr} on the GPU
    (the body of AudioDataset.__getitem__, audio_dataset.py:55-61, for a whole batch)."""

    def __init__(self, opt, device=None):
        self.lr_sampling_rate = int(opt.lr_sampling_rate)
        self.hr_sampling_rate = int(opt.hr_sampling_rate)
        self.segment_length = int(opt.segment_length)
        self.device = torch.device(device if device is not None else 'cuda')

    def __call__(self, batch):
        raw = batch['raw'].to(self.device, non_blocking=True).float()
        lens = batch['raw_len'].to(self.device).long()
        rates = batch['rate'].tolist() if torch.is_tensor(batch['rate']) else list(batch['rate'])
        hr = torch.empty((raw.shape[0], self.segment_length), dtype=torch.float32, device=self.device)
        lr = torch.empty_like(hr)
        for rate in sorted(set(rates)):                               # one launch chain per distinct source rate
            rows = torch.tensor([i for i, r in enumerate(rates) if r == rate], device=self.device)
            x, n = raw[rows], lens[rows]
            h, _ = _masked_resample(x, n, rate, self.hr_sampling_rate)
            l, nl = _masked_resample(x, n, rate, self.lr_sampling_rate)
            l, _ = _masked_resample(l, nl, self.lr_sampling_rate, self.hr_sampling_rate)
            hr[rows] = _fit(h, self.segment_length)
            lr[rows] = _fit(l, self.segment_length)
        out = {k: v for k, v in batch.items() if k not in ('raw', 'raw_len', 'rate')}
        out.update({'image': hr, 'label': lr})
        return out


class AudioDataset(BaseDataset):
    """Training segments (audio_dataset.py:10-88).  `__getitem__` is worker-safe (I/O only) and returns the raw
    segment; `resolve(item_or_batch)` / `GpuFeeder` turn it into the reference's {'image': hr, 'label': lr}."""

    def __init__(self, opt) -> None:
        super().__init__()
        self.lr_sampling_rate = opt.lr_sampling_rate
        self.hr_sampling_rate = opt.hr_sampling_rate
        self.segment_length = opt.segment_length
        self.n_fft = opt.n_fft
        self.hop_length = opt.hop_length
        self.win_length = opt.win_length
        self.audio_file = self.get_files(opt.dataroot)
        self.center = opt.center
        self.opt = opt
        torch.manual_seed(opt.seed)

    def __len__(self):
        return len(self.audio_file)

    def name(self):
        return 'AudioMDCTSpectrogramDataset'

    def readaudio(self, file_path):
        metadata = wavio.info(file_path)
        max_audio_start = metadata.num_frames - self.segment_length
        if max_audio_start > 0:
            offset = torch.randint(low=0, high=max_audio_start, size=(1,)).item()
            return wavio.load(file_path, frame_offset=offset, num_frames=self.segment_length)
        print("Warning: %s is shorter than segment_length" % file_path, metadata.num_frames)
        return wavio.load(file_path)

    def __getitem__(self, idx):
        file_path = self.audio_file[idx]
        try:
            waveform, rate = self.readaudio(file_path)
        except (OSError, ValueError):                                 # try the next files until one loads (:44-53)
            i = 1
            while True:
                print('Load failed!')
                file_path = self.audio_file[(idx + i) % len(self.audio_file)]
                try:
                    waveform, rate = self.readaudio(file_path)
                    break
                except (OSError, ValueError):
                    i += 1
                    if i > len(self.audio_file):
                        raise
        mono = waveform[0]
        n = min(mono.numel(), self.segment_length)
        raw = torch.zeros(self.segment_length, dtype=torch.float32)
        raw[:n] = mono[:n]
        return {'raw': raw, 'raw_len': n, 'rate': rate, 'inst': 0, 'feat': 0, 'path': file_path}

    def resolve(self, item, device=None):
        """One item (or a collated batch) through the GPU stage: the dict the reference's __getitem__ returns."""
        single = item['raw'].dim() == 1
        batch = dict(item)
        if single:
            batch.update(raw=item['raw'][None], raw_len=torch.tensor([item['raw_len']]), rate=[item['rate']])
        out = GpuFeeder(self.opt, device)(batch)
        if single:
            out['image'], out['label'] = out['image'][0], out['label'][0]
        return out

    def get_files(self, file_path):
        """Directory -> every .wav below it; otherwise a csv whose cells are paths relative to the csv's folder
        (audio_dataset.py:64-79; the reference's extension test accepts anything, this build decodes RIFF/WAVE only)."""
        if os.path.isdir(file_path):
            found = [os.path.join(folder, name) for folder, _, names in os.walk(file_path, topdown=False)
                     for name in names if name.lower().endswith(".wav")]
            print("Searching for audio file")
        else:
            base = os.path.dirname(file_path)
            with open(file_path, newline="") as fh:
                found = [os.path.join(base, cell) for line in csv.reader(fh) for cell in line]
            print("Using csv file list")
        print(len(found))
        return found

    def seg_pad_audio(self, waveform):
        """[C, T] -> first channel cropped to segment_length (1-D), or the [C, segment_length] right-padded tensor."""
        missing = self.segment_length - waveform.size(1)
        return waveform[0][:self.segment_length] if missing <= 0 else F.pad(waveform, (0, missing))


class AudioTestDataset(BaseDataset):
    """Whole-file inference input (audio_dataset.py:89-135): `raw_audio`, `lr_audio` [1, T] and `seg_audio`
    [segments, segment_length] live on the GPU; the LR round trip runs on csrc/resample.hip."""

    def __init__(self, opt, device=None) -> None:
        super().__init__()
        self.lr_sampling_rate = opt.lr_sampling_rate
        self.hr_sampling_rate = opt.hr_sampling_rate
        self.segment_length = opt.segment_length
        self.n_fft = opt.n_fft
        self.hop_length = opt.hop_length
        self.win_length = opt.win_length
        self.center = opt.center
        self.dataroot = opt.dataroot
        self.device = torch.device(device if device is not None else 'cuda')
        raw, self.in_sampling_rate = wavio.load(self.dataroot)
        self.raw_audio = raw.to(self.device)
        self.audio_len = self.raw_audio.size(-1)
        print("Audio length:", self.audio_len)
        if getattr(opt, 'is_lr_input', False):
            self.lr_audio = resample(self.raw_audio, self.in_sampling_rate, self.hr_sampling_rate)
        else:
            self.lr_audio = resample(self.raw_audio, self.in_sampling_rate, self.lr_sampling_rate)
            self.lr_audio = resample(self.lr_audio, self.lr_sampling_rate, self.hr_sampling_rate)
        self.seg_audio = self.seg_pad_audio(self.lr_audio)

    def __len__(self):
        return self.seg_audio.size(0)

    def name(self):
        return 'AudioMDCTSpectrogramTestDataset'

    def __getitem__(self, idx):
        return {'image': torch.empty(1), 'label': self.seg_audio[idx, :].squeeze(0), 'inst': torch.empty(1),
                'feat': torch.empty(1), 'path': self.dataroot}

    def seg_pad_audio(self, audio):
        """[1, T] or [T] -> [ceil(T / segment_length), segment_length], zero padded at the end."""
        audio = audio.squeeze(0)
        segments = max(1, int(math.ceil(len(audio) / self.segment_length)))
        audio = F.pad(audio, (0, segments * self.segment_length - len(audio)))
        return audio.view(segments, self.segment_length)

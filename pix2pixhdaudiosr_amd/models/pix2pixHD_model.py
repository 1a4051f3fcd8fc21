"""Pix2PixHDModel with the reference's public surface (models/pix2pixHD_model.py), driving the HIP hot path:
MDCT4 -> dB/sign encoding -> generator -> 3x multiscale discriminator -> LSGAN + feature-matching losses.

What is the same as the reference: constructor-less ``initialize(opt)``; ``forward(lr_audio, inst, hr_audio,
feat, infer)`` returning ``[filtered losses, sr or None]``; ``inference``; ``encode_input`` (8-tuple);
``to_spectro`` / ``denormalize`` / ``to_audio``; ``loss_names``; ``optimizer_G`` / ``optimizer_D``;
``save`` / ``update_fixed_params`` / ``update_learning_rate``; checkpoint file names and state_dict keys.

Deliberate differences:
  * the transform defaults to MDCT4/IMDCT4 (n_fft/2 bins -- the 512x256 geometry of BASELINE.json); the shipped
    reference hard-codes MDCT2 (pix2pixHD_model.py:37-40) and invites the swap in README.md:133; ``opt.mdct_type =
    'mdct2'`` selects the reference's own MDCT2/IMDCT2 (n_fft bins) and enables to_frames / --use_match_loss;
  * MDCT output is fp32 (the reference's complex128 twiddles make it fp64, which its own fp32 convs reject);
  * every to_spectro configuration (explicit two-channel or single-channel encoding, all mask / phase modes);
    LSGAN, no VGG / hifigan / time-domain discriminator / feature encoder;
  * the D weight gradients of the G-loss pass are not computed (train.py:176 zeroes them unread);
  * the per-step device->host copies of pix2pixHD_model.py:418-428 are deferred to get_current_visuals();
  * ``--fp16`` selects bf16 MFMA compute with fp32 master weights instead of fp16 autocast + GradScaler;
  * ``.module`` returns the model itself (create_model never wraps in DataParallel, see models.py).
"""
import numpy as np
import os

import torch

from .. import _lib, _ops
from ..optim import FlatAdam
from ..util.util import kbdwin
from . import networks
from .base_model import BaseModel


def _default_mdct_type():
    """'mdct4' (BASELINE's 512x256 geometry) unless the environment says otherwise: the zero-edit launcher
    (pix2pixhdaudiosr_amd.dropin) sets P2PHD_MDCT_TYPE=mdct2, the transform the shipped reference hard-codes and its eval
    code inverts (train.py:58-60)."""
    return os.environ.get('P2PHD_MDCT_TYPE', 'mdct4')
from .mdct import MDCT4, IMDCT4, MDCT2, IMDCT2


def _opt(opt, name, default):
    return getattr(opt, name, default)


class Pix2PixHDModel(BaseModel):
    def name(self):
        return 'Pix2PixHDModel'

    @property
    def module(self):
        return self

    def init_loss_filter(self, use_gan_feat_loss, use_vgg_loss, use_match_loss, use_time_loss):
        flags = (True, use_gan_feat_loss, use_vgg_loss, use_match_loss, use_time_loss, use_time_loss, use_time_loss, True, True)

        def loss_filter(g_gan, g_gan_feat, g_vgg, g_mat, g_gan_t, d_real_t, d_fake_t, d_real, d_fake):
            return [l for (l, f) in zip((g_gan, g_gan_feat, g_vgg, g_mat, g_gan_t, d_real_t, d_fake_t, d_real, d_fake), flags) if f]
        return loss_filter

    def _check_supported(self, opt):
        unsupported = []
        if _opt(opt, 'mask_mode', None) not in (None, 'mode0', 'mode1', 'mode2'):
            unsupported.append("mask_mode must be None, 'mode0', 'mode1' or 'mode2'")
        if _opt(opt, 'phase_encoding_mode', None) not in (None, 'uni_dist', 'norm_dist', 'norm_dist2', 'scale'):
            unsupported.append("phase_encoding_mode must be None, 'uni_dist', 'norm_dist', 'norm_dist2' or 'scale'")
        if not _opt(opt, 'explicit_encoding', False) and _opt(opt, 'use_match_loss', False):
            unsupported.append("--use_match_loss needs explicit_encoding (to_frames returns None without it, pix2pixHD_model.py:255-256)")
        for flag in ('use_hifigan_D', 'use_time_D', 'instance_feat', 'label_feat'):
            if _opt(opt, flag, False):
                unsupported.append("--%s is outside the HIP hot path" % flag)
        if not _opt(opt, 'no_vgg_loss', True):
            unsupported.append("VGG loss (deprecated in the reference) needs --no_vgg_loss")
        if not _opt(opt, 'no_instance', True):
            unsupported.append("instance maps are deprecated: use --no_instance")
        if _opt(opt, 'no_lsgan', False):
            unsupported.append("only the LSGAN criterion is implemented")
        if _opt(opt, 'label_nc', 0) != 0:
            unsupported.append("label_nc must be 0 (audio)")
        if _opt(opt, 'pool_size', 0) != 0:
            unsupported.append("pool_size must be 0 (the reference default; ImagePool is a no-op then)")
        if _opt(opt, 'use_match_loss', False) and _opt(opt, 'mdct_type', _default_mdct_type()) != 'mdct2':
            unsupported.append("--use_match_loss compares DCT frames: it needs mdct_type='mdct2'")
        if _opt(opt, 'mdct_type', _default_mdct_type()) not in ('mdct4', 'mdct2'):
            unsupported.append("mdct_type must be 'mdct4' or 'mdct2'")
        if unsupported:
            raise NotImplementedError("Pix2PixHDModel (HIP path): " + "; ".join(unsupported))

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        self._check_supported(opt)
        if not torch.cuda.is_available() or len(self.gpu_ids) == 0:
            raise RuntimeError("this model runs on the MI355X HIP kernels only; pass gpu_ids=[0] (no CPU path)")
        self.isTrain = opt.isTrain
        self.use_features = False
        self.gen_features = False
        # --fp16 = 16-bit activation storage with fp32 accumulation and fp32 master weights.  Default bf16 (the benchmarked mode,
        # BASELINE configs[1]); opt.fp16_storage = True: IEEE fp16, the reference's actual autocast type (train.py:62-67), served
        # by the fp16 build of the library (libp2phd_hip_f16.so) with a device-resident loss scaler (optim.DeviceGradScaler)
        self.compute_dtype = torch.float32
        if _opt(opt, 'fp16', False):
            self.compute_dtype = torch.float16 if _opt(opt, 'fp16_storage', False) else torch.bfloat16
        input_nc = opt.input_nc

        ##### transform
        self.up_ratio = opt.hr_sampling_rate / opt.lr_sampling_rate
        self.window = kbdwin(opt.win_length).to(self.device)
        kw = dict(n_fft=opt.n_fft, hop_length=opt.hop_length, win_length=opt.win_length, window=self.window, device=self.device)
        # mdct_type 'mdct4' (default: n_fft/2 bins, the BASELINE 512x256 geometry) or 'mdct2' (what the shipped
        # reference hard-codes, pix2pixHD_model.py:37-40: n_fft bins through DCT_2N_native / IDCT_2N_native)
        self.mdct_type = _opt(opt, 'mdct_type', _default_mdct_type())
        if self.mdct_type == 'mdct2':
            from ..dct.dct_native import DCT_2N_native, IDCT_2N_native
            self._dct, self._idct = DCT_2N_native(), IDCT_2N_native()
            self._mdct = MDCT2(dct_op=self._dct, **kw)
            self._imdct = IMDCT2(idct_op=self._idct, **kw)
        else:
            self._mdct = MDCT4(**kw)
            self._imdct = IMDCT4(**kw)

        ##### networks
        verbose = _opt(opt, 'verbose', False)
        self.netG = networks.define_G(input_nc, opt.output_nc, opt.ngf, opt.netG, opt.n_downsample_global,
                                      opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local, opt.norm,
                                      gpu_ids=self.gpu_ids, dtype=self.compute_dtype, verbose=verbose)
        if self.isTrain:
            self.netD = networks.define_D(input_nc + opt.output_nc, opt.ndf, opt.n_layers_D, opt.norm, False, opt.num_D,
                                          not opt.no_ganFeat_loss, gpu_ids=self.gpu_ids, dtype=self.compute_dtype,
                                          verbose=verbose)
        # opt.fp8: e4m3 forward of the wide layers on top of bf16 compute (BASELINE configs[4])
        self.fp8_layers = 0
        if _opt(opt, 'fp8', False):
            if self.compute_dtype != torch.bfloat16:
                raise NotImplementedError("--fp8 needs bf16 compute (--fp16)")
            self.fp8_layers = networks.enable_fp8(self.netG)
            if self.isTrain:
                self.fp8_layers += networks.enable_fp8(self.netD)
        if verbose:
            print('---------- Networks initialized -------------')

        if not self.isTrain or _opt(opt, 'continue_train', False) or _opt(opt, 'load_pretrain', ''):
            pretrained_path = '' if not self.isTrain else opt.load_pretrain
            self.load_network(self.netG, 'G', opt.which_epoch, pretrained_path)
            if self.isTrain:
                self.load_network(self.netD, 'D', opt.which_epoch, pretrained_path)

        if self.isTrain:
            self.old_lr = opt.lr
            self.loss_filter = self.init_loss_filter(not opt.no_ganFeat_loss, False, bool(_opt(opt, 'use_match_loss', False)), False)
            self.criterionGAN = networks.GANLoss(use_lsgan=True, tensor=self.Tensor)
            self.criterionFeat = networks.FeatLoss()
            self.loss_names = self.loss_filter('G_GAN', 'G_GAN_Feat', 'G_VGG', 'G_mat', 'G_GAN_t', 'D_real_t', 'D_fake_t', 'D_real', 'D_fake')

            if _opt(opt, 'niter_fix_global', 0) > 0:
                prefix = 'model' + str(opt.n_local_enhancers)
                params = [v for k, v in self.netG.named_parameters() if k.startswith(prefix)]
                print('------------- Only training the local enhancer network (for %d epochs) ------------' % opt.niter_fix_global)
            else:
                params = list(self.netG.parameters())
            if verbose:
                print('Total number of parameters of G: %d' % (sum([param.numel() for param in params])))
            self.optimizer_G = FlatAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999))
            params = list(self.netD.parameters())
            if verbose:
                print('Total number of parameters of D: %d' % (sum([param.numel() for param in params])))
            self.optimizer_D = FlatAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999))
            self._attach_scaler()
            if self.scaler is not None and _opt(opt, 'continue_train', False):
                self.scaler.load_file(os.path.join(self.save_dir, '%s_scaler.pth' % opt.which_epoch))
        self._visual = None

    # ------------------------------------------------------------------------------------------
    # spectrogram codec (HIP: csrc/spectro.hip)
    # ------------------------------------------------------------------------------------------
    def to_spectro(self, audio, mask=False, noise=None, phase_noise=None, noise_sign=None, _spec=None):
        """audio [B,T] -> (log_spectro [B,C,bins,frames] in [0,1], pha [B,1,bins,frames], norm dict); C = 2 with
        explicit_encoding, else 1 (pix2pixHD_model.py:142-227).  The random tensors the reference draws inside can be
        handed in (parity tests): ``noise`` [B,C,mask_rows,frames] (torch.randn of :202), ``noise_sign`` (+-1, the randint of
        :215 for mask_mode 'mode1'), ``phase_noise`` [B,1,bins,frames] (the rand / randn of :180-188)."""
        spec = _spec if _spec is not None else self._mdct(audio.to(self.device))   # [B, frames, bins] f32
        B, Fr, M = spec.shape
        L = _lib.lib()
        explicit = bool(_opt(self.opt, 'explicit_encoding', False))
        C = 2 if explicit else 1
        pem = None if explicit else _opt(self.opt, 'phase_encoding_mode', None)
        if pem in ('uni_dist', 'norm_dist', 'norm_dist2') and phase_noise is None:      # drawn first, as the reference does
            phase_noise = (torch.rand if pem == 'uni_dist' else torch.randn)((B, 1, M, Fr), device=spec.device)
        log_spectro = torch.empty((B, C, M, Fr), dtype=torch.float32, device=spec.device)
        pha = torch.empty((B, 1, M, Fr), dtype=torch.float32, device=spec.device)
        norm8 = torch.zeros(8, dtype=torch.float32, device=spec.device)
        partials = torch.empty(L.p2phd_spectro_partials_floats(B, Fr, M), dtype=torch.float32, device=spec.device)
        mask_rows = 0
        mode = _opt(self.opt, 'mask_mode', None)
        if mask:
            mask_rows = int(M * (1 - 1 / self.up_ratio))                    # pix2pixHD_model.py:199
            if mode in ('mode0', 'mode1', 'mode2'):
                if noise is None:
                    noise = torch.randn(B, C, mask_rows, Fr, device=spec.device)
                noise = noise.to(spec.device).float().contiguous()
                assert tuple(noise.shape) == (B, C, mask_rows, Fr)
                if mode == 'mode1':
                    if noise_sign is None:
                        noise_sign = 2 * torch.randint(low=0, high=2, size=noise.size(), device=spec.device) - 1
                    noise_sign = noise_sign.to(spec.device).float().contiguous()
                    assert noise_sign.shape == noise.shape
                else:
                    noise_sign = None
            else:
                noise = noise_sign = None
        else:
            noise = noise_sign = None
        _lib.check(L.p2phd_spectro_encode_ex(_lib.ptr(spec), B, Fr, M, C, float(self.opt.alpha), float(self.opt.min_value),
                                             mask_rows, {'mode0': 0, 'mode1': 1}.get(mode, 2), _lib.ptr(noise),
                                             _lib.ptr(noise_sign), _lib.ptr(log_spectro), _lib.ptr(pha), _lib.ptr(norm8),
                                             _lib.ptr(partials), _lib.stream_ptr()), "spectro_encode")
        if pem is not None:                                                 # :178-191 (single-channel encoding only)
            if pem == 'scale':
                pha = pha * 0.5
            else:
                pn = phase_noise.to(spec.device).float()
                if pem == 'norm_dist':
                    pn = (pn - pn.min()) / (pn.max() - pn.min())
                elif pem == 'norm_dist2':
                    pn = pn.abs()
                pha = pha * pn
        norm = {'min': norm8[0], 'max': norm8[1], 'mean': norm8[2], 'std': norm8[3], 'frames': None, '_minmax': norm8[:2]}
        return log_spectro, pha, norm

    def _minmax(self, norm_param):
        mm = norm_param.get('_minmax')
        if mm is None:
            mm = torch.stack([torch.as_tensor(norm_param['min']).float().reshape(()),
                              torch.as_tensor(norm_param['max']).float().reshape(())]).to(self.device)
        return mm.contiguous()

    def denormalize(self, log_spectro, norm_param):
        mm = self._minmax(norm_param)
        s = torch.abs(log_spectro) * (mm[1] - mm[0]) + mm[0]
        return 10.0 * torch.pow(10.0, s * 0.05) - self.opt.min_value        # DB_to_amplitude(s, 10, 0.5) - min_value

    def to_audio(self, log_spectro, norm_param, pha=None, pseudo_pha=None):
        x = log_spectro.to(self.device).float().contiguous()
        B, _, M, Fr = x.shape
        spec = torch.empty((B, Fr, M), dtype=torch.float32, device=x.device)
        if not _opt(self.opt, 'explicit_encoding', False):
            # single-channel encoding (pix2pixHD_model.py:238-249): amplitude times pha on the bins the low-rate input
            # carries and times a random sign above (``pseudo_pha`` replaces the randint of :240); with up_ratio <= 1 the
            # reference never applies the sign at all
            if self.up_ratio > 1:
                keep = int(M * (1 / self.up_ratio))
                if pseudo_pha is None:
                    pseudo_pha = 2 * torch.randint(low=0, high=2, size=pha.size(), device=x.device) - 1
                sign = torch.cat((pha.to(x.device).float()[..., :keep, :], pseudo_pha.to(x.device).float()[..., keep:, :]), dim=-2)
            else:
                sign = torch.ones((B, 1, M, Fr), dtype=torch.float32, device=x.device)
            sign = sign.reshape(B, M, Fr).contiguous()
            _lib.check(_lib.lib().p2phd_spectro_decode_signed(_lib.ptr(x), _lib.ptr(sign), _lib.ptr(self._minmax(norm_param)), B, Fr,
                                                              M, 1, M, float(self.opt.min_value), 1.0, _lib.ptr(spec),
                                                              _lib.stream_ptr()), "spectro_decode")
            return np.sqrt(self.up_ratio - 1) * self._imdct(spec)
        _lib.check(_lib.lib().p2phd_spectro_decode(_lib.ptr(x), _lib.ptr(self._minmax(norm_param)), B, Fr, M,
                                                   float(self.opt.alpha), float(self.opt.min_value), _lib.ptr(spec),
                                                   _lib.stream_ptr()), "spectro_decode")
        return np.sqrt(self.up_ratio - 1) * self._imdct(spec)

    def to_frames(self, log_spectro, norm_param):
        """Un-windowed time-domain frames of an MDCT2 spectrogram (pix2pixHD_model.py:251-258): differentiable, the
        IDCT runs on the HIP kernel; the small dB decode in front of it is plain tensor arithmetic."""
        if not _opt(self.opt, 'explicit_encoding', False):
            return None                                                     # pix2pixHD_model.py:255-256
        if self.mdct_type != 'mdct2':
            raise NotImplementedError("to_frames needs mdct_type='mdct2' (frames are IDCT_2N_native rows)")
        spectro = self.denormalize(log_spectro, norm_param)
        spectro = (spectro[..., 0, :, :] - spectro[..., 1, :, :]) / (2 * self.opt.alpha - 1)
        return self._idct(spectro.permute(0, 2, 1).contiguous())

    def encode_input(self, lr_audio, inst_map=None, hr_audio=None, feat_map=None, noise=None):
        with torch.no_grad():
            hr_spec = lr_spec = None
            if hr_audio is not None and tuple(hr_audio.shape) == tuple(lr_audio.shape) and hr_audio.dim() == 2:
                # both clips through ONE transform launch (2B rows: the launch ramp of a 12 us kernel is paid once); the
                # codec then normalises each half on its own, as the reference does (pix2pixHD_model.py:302-320)
                both = self._mdct(torch.cat((hr_audio.to(self.device), lr_audio.to(self.device)), dim=0), _dim0=hr_audio.shape[0])
                hr_spec, lr_spec = both[:hr_audio.shape[0]], both[hr_audio.shape[0]:]
            if hr_audio is not None:
                hr_spectro, hr_pha, hr_norm_param = self.to_spectro(hr_audio, mask=False, _spec=hr_spec)
            else:
                hr_spectro = hr_pha = hr_norm_param = None
            lr_spectro, lr_pha, lr_norm_param = self.to_spectro(lr_audio, mask=bool(_opt(self.opt, 'mask', False)), noise=noise,
                                                                _spec=lr_spec)
        return lr_spectro, lr_pha, hr_spectro, hr_pha, feat_map, inst_map, hr_norm_param, lr_norm_param

    # ------------------------------------------------------------------------------------------
    # losses on physical tensors
    # ------------------------------------------------------------------------------------------
    def _D(self, lr_spectro, other):
        """netD on cat(lr, other) -> list[num_D] of list of (physical tensor, channels)."""
        # exclusive: intermediate features leave this module only through _losses, whose feature-matching terms park
        # their gradient (l1_loss(..., park=True)) and whose real-side features are detached
        return self.netD.forward_physical(_ops.ToPhysical.apply(self.compute_dtype, lr_spectro, other), exclusive=True)

    @staticmethod
    def _gan(pred, target):
        acc = _ops.LossAcc(pred[0][-1][0].device)                  # networks.py:100-110: sum over scales, one accumulator
        for scale in pred:
            t, c = scale[-1]
            acc.mse_const(t, c, target)
        return acc.total()

    def discriminate_F(self, input_label, test_image, use_pool=False):
        return self.netD.forward(torch.cat((input_label, test_image.detach()), dim=1))

    def _d_pair(self):
        """Run D(real) and D(fake) of the training step as one batch (default; opt.d_pair = False or P2PHD_DPAIR=0 keeps
        the two passes of rounds 1-2 for A/B runs).  Needs intermediate features only through the parked losses."""
        import os
        return bool(_opt(self.opt, 'd_pair', True)) and os.environ.get("P2PHD_DPAIR", "1") != "0"

    def _fake_half(self):
        """Context of the generator-loss backward: D's Functions (batch 2B) process the fake half only."""
        import contextlib
        n = getattr(self, '_pair_batch', None)
        return _ops.backward_on_samples(n, n // 2, n) if n else contextlib.nullcontext()

    def _losses(self, lr_audio, hr_audio, noise, share_fake_pass):
        """All loss terms of one step.  share_fake_pass=False is the reference's schedule (pix2pixHD_model.py:331-415):
        D(fake.detach()), D(real), D(fake).  share_fake_pass=True runs D on the generated spectrogram ONCE: the detached
        and the attached pass compute identical values, only their backward differs, so `train_step` walks the one
        retained graph twice (G loss without D weight gradients, then D loss restricted to D's parameters)."""
        _ops.begin_step(self.device)                               # one memset for every statistics / loss accumulator
        lr_spectro, lr_pha, hr_spectro, hr_pha, _, _, hr_norm_param, lr_norm_param = \
            self.encode_input(lr_audio, None, hr_audio, None, noise=noise)

        # cut points of the staged generator backward (data parallel: one gradient bucket per stage, see _g_stages)
        _, cut_after, _ = self._bucket_plan()
        self._cuts = [] if cut_after else None
        if cut_after:
            sr_phys = self.netG.forward_physical(self.netG.input_physical(lr_spectro), cuts=self._cuts,
                                                 cut_after=set(cut_after))
        else:
            sr_phys = self.netG.forward_physical(self.netG.input_physical(lr_spectro))
        sr_result = _ops.FromPhysical.apply(sr_phys, self.opt.output_nc)

        pair = share_fake_pass and self._d_pair()
        if pair:
            # D(real) and D(fake) as ONE batch of 2B (real half first): same weights, one launch per layer instead of two,
            # fuller tile rounds (e.g. 1122 tiles instead of 2 x 561 on the 256 -> 512 layer).  InstanceNorm is per sample, so
            # every value equals the two-pass result.  The generator-loss backward then runs on the fake half only
            # (_ops.backward_on_samples in _g_stages), the discriminator-loss backward on the whole batch.
            B = int(lr_spectro.shape[0])
            pred = self.netD.forward_physical(_ops.to_physical_pair(self.compute_dtype, lr_spectro, hr_spectro, sr_result),
                                              exclusive=True)
            self._pair_batch = 2 * B
            a_real, a_fake, a_gan = (_ops.LossAcc(self.device) for _ in range(3))
            for scale in pred:
                t, c = scale[-1]
                a_real.mse_const(t, c, 1.0, rows=(0, B))
                a_fake.mse_const(t, c, 0.0, rows=(B, 2 * B))
                a_gan.mse_const(t, c, 1.0, rows=(B, 2 * B))
            loss_D_real, loss_D_fake, loss_G_GAN = a_real.total(), a_fake.total(), a_gan.total()
        elif share_fake_pass:
            pred_real = self._D(lr_spectro, hr_spectro)
            pred_fake = self._D(lr_spectro, sr_result)
            loss_D_fake = self._gan(pred_fake, 0.0)
        else:
            # fake detection (detached: no gradient to G), real detection
            pred_fake_pool = self._D(lr_spectro, sr_result.detach())
            loss_D_fake = self._gan(pred_fake_pool, 0.0)
            pred_real = self._D(lr_spectro, hr_spectro)
            # GAN loss through D into G; D's weight gradients of this pass are never used (train.py:176)
            with _ops.no_weight_grad():
                pred_fake = self._D(lr_spectro, sr_result)
        if not pair:
            self._pair_batch = None
            loss_D_real = self._gan(pred_real, 1.0)
            loss_G_GAN = self._gan(pred_fake, 1.0)

        loss_G_GAN_Feat = 0
        if not self.opt.no_ganFeat_loss:
            feat_weights = 4.0 / (self.opt.n_layers_D + 1)
            D_weights = 1.0 / self.opt.num_D
            w_feat = D_weights * feat_weights * self.opt.lambda_feat
            a_feat = _ops.LossAcc(self.device)
            for i in range(self.opt.num_D):
                if pair:
                    for j in range(len(pred[i]) - 1):
                        t, c = pred[i][j]
                        a_feat.l1_halves(t, c, w_feat, park=True)
                    continue
                for j in range(len(pred_fake[i]) - 1):
                    (a, c), (b, _) = pred_fake[i][j], pred_real[i][j]
                    a_feat.l1(a, b, c, w_feat, park=True)
            loss_G_GAN_Feat = a_feat.total()

        # TDAC frame-matching loss (pix2pixHD_model.py:408-415): the second half of frame t and the first half of frame
        # t+1, each under its window half, must coincide
        loss_G_match = 0
        if _opt(self.opt, 'use_match_loss', False):
            half = self.opt.win_length // 2
            sr_frames = self.to_frames(sr_result, lr_norm_param)
            a = sr_frames[..., :-1, half:] * self.window[:half]
            b = sr_frames[..., 1:, :half] * self.window[half:]
            loss_G_match = torch.nn.functional.mse_loss(a, b) * self.opt.lambda_mat

        # visuals are fetched lazily (no device->host copy in the step)
        self._visual = (lr_spectro, sr_result.detach(), hr_spectro, hr_pha)
        _ops.end_arena(self.device)
        return self.loss_filter(loss_G_GAN, loss_G_GAN_Feat, 0, loss_G_match, 0, 0, 0, loss_D_real, loss_D_fake), sr_result

    def forward(self, lr_audio, inst, hr_audio, feat, infer=False, noise=None):
        losses, sr_result = self._losses(lr_audio, hr_audio, noise, share_fake_pass=False)
        # state of the staged step (cut activations, combined losses) belongs to train_step only: the reference-style
        # forward() / backward() path never consumes it, and it would keep the autograd graph alive until the next forward
        self._cuts = None
        self._loss_G = self._loss_D = None
        return [losses, None if not infer else sr_result]

    def inference(self, lr_audio, inst, noise=None):
        lr_spectro, lr_pha, _, _, _, _, _, lr_norm_param = self.encode_input(lr_audio, inst, None, noise=noise)
        with torch.no_grad():
            sr_spectro = self.netG.forward(lr_spectro)
        return sr_spectro, lr_pha, lr_norm_param, lr_spectro

    # ------------------------------------------------------------------------------------------
    # one full optimisation step (train.py:148-184) with the all-reduce of the G gradients overlapped
    # with the D backward pass; result-identical to calling backward/step in train.py's order
    # ------------------------------------------------------------------------------------------
    def _bucket_plan(self):
        """(n_buckets, cut_after, offsets): where the generator backward is cut into stages and the element offset in
        optimizer_G's flat buffers at which each cut's later layers start.  `opt.grad_buckets` overrides the default
        (4 with data parallelism, 1 without).  Needs optimizer_G to own all of netG's parameters in module order."""
        if not self.isTrain:
            return 1, [], []
        optG = self.optimizer_G
        n = int(_opt(self.opt, 'grad_buckets', 4 if optG.world_size > 1 else 1))
        plan = getattr(self, '_bucket_plan_cache', None)
        if plan is None or plan[0] != (n, id(optG)):
            cut_after, offs = [], []
            whole = len(optG._params) == len(list(self.netG.parameters()))
            if n > 1 and whole and hasattr(self.netG, 'bucket_plan'):
                cut_after, first = self.netG.bucket_plan(n)
                offs = [optG.param_offset(i) for i in first]
            plan = ((n, id(optG)), cut_after, offs)
            self._bucket_plan_cache = plan
        return plan[0][0], plan[1], plan[2]

    def _attach_scaler(self):
        """fp16 storage: one device-resident loss scaler for both optimisers (train.py:62-67); none otherwise."""
        from ..optim import DeviceGradScaler
        self.scaler = None
        if self.compute_dtype == torch.float16:
            self.scaler = DeviceGradScaler(self.device, init_scale=float(_opt(self.opt, 'loss_scale', 65536.0)))
        for idx, o in enumerate((self.optimizer_G, self.optimizer_D)):
            o.scaler, o.scaler_index = self.scaler, idx

    def _phase_a_forward(self, lr_audio, hr_audio, noise=None):
        """Forward of G and D, all losses, zeroed gradient buffers.  Returns the loss dict; the generator backward is
        then run stage by stage with `_g_stages()`."""
        losses, _ = self._losses(lr_audio, hr_audio, noise, share_fake_pass=True)
        ld = dict(zip(self.loss_names, losses))
        self._loss_D = (ld['D_fake'] + ld['D_real']) * 0.5
        self._loss_G = ld['G_GAN'] + ld.get('G_GAN_Feat', 0) + ld.get('G_mat', 0)
        if self.scaler is not None:                                # scaler.scale(loss).backward() of train.py:165-181
            self._loss_D, self._loss_G = self.scaler.scale(self._loss_D), self.scaler.scale(self._loss_G)
        self.optimizer_G.zero_grad(lazy=True)                       # (first weight gradient of the step overwrites: no memset)
        self.optimizer_D.zero_grad(lazy=True)
        self.optimizer_G.bucket_log = []
        self.optimizer_D.bucket_log = []
        return ld

    def _g_stages(self):
        """The backward of the generator loss as a list of (run, (start, stop)) stages, last layers first.  Running a
        stage computes the weight gradients of its layers straight into optimizer_G.flat_g[start:stop] (a contiguous
        range: parameters are laid out in execution order), so the caller can start that range's all-reduce while the
        next stage runs (bucketed exchange overlapped with the backward).  Only G parameters receive
        gradients (D's convs skip their weight-gradient kernels in this pass); the graph through D(fake) is kept for
        the D loss.  One stage = today's single backward."""
        optG, optD = self.optimizer_G, self.optimizer_D
        cuts, loss_G = self._cuts, self._loss_G
        _, _, offs = self._bucket_plan()
        total = optG._total
        if not cuts:
            def whole():
                with _ops.backward_without_weight_grads(optD._params), self._fake_half():
                    loss_G.backward(inputs=list(optG._params), retain_graph=True)
            return [(whole, (0, total))]
        state = {}
        stages = []

        anchors = list(getattr(self.netG, 'staged_head_anchors', lambda: [])())

        def head():
            with _ops.backward_without_weight_grads(optD._params), self._fake_half():
                # (anchors: parameters of a branch parallel to the last cut -- the LocalEnhancer's head; their gradient
                # arrives in the flat buffer like every other one, autograd itself gets None for them)
                state['g'] = torch.autograd.grad(loss_G, [cuts[-1]] + anchors, retain_graph=True, allow_unused=True)[0]
        stages.append((head, (offs[-1], total)))
        for i in range(len(cuts) - 2, -1, -1):
            def mid(i=i):
                (state['g'],) = torch.autograd.grad(cuts[i + 1], [cuts[i]], grad_outputs=state['g'], retain_graph=True)
            stages.append((mid, (offs[i], offs[i + 1])))
        first_params = [p for p, o in zip(optG._params, optG._offs) if o < offs[0]]

        def tail():
            cuts[0].backward(gradient=state['g'], inputs=first_params, retain_graph=True)
            state.clear()
        stages.append((tail, (0, offs[0])))
        return stages

    def _phase_a(self, lr_audio, hr_audio, noise=None):
        """Forward, losses and the whole generator backward, with each gradient bucket's all-reduce started as soon as
        its stage is done (no-ops on one GPU)."""
        ld = self._phase_a_forward(lr_audio, hr_audio, noise)
        for run, (a, b) in self._g_stages():
            run()
            self.optimizer_G.reduce_range_async(a, b)
        self._loss_G = None
        return ld

    def _phase_b(self):
        """Backward of the discriminator loss through the graph phase A kept."""
        loss_D, self._loss_D = self._loss_D, None
        firsts = [self.netD._scale_steps(d)[0][0].spec for d in range(self.opt.num_D)]
        with _ops.backward_without_input_grads(firsts):   # no gradient towards the generator in this pass
            loss_D.backward(inputs=list(self.optimizer_D._params))

    # opt.comm_cus = N > 0 (data parallel only makes sense): the step runs on a stream whose CU mask leaves N CUs to the
    # RCCL kernels of the gradient exchange (parallel_state.masked_compute_stream)
    def _on_step_stream(self, fn, *args, **kw):
        n = int(_opt(self.opt, 'comm_cus', 0) or 0)
        if n <= 0:
            return fn(*args, **kw)
        st = getattr(self, '_step_stream', None)
        if st is None or getattr(st, '_p2phd_free_cus', None) != n:
            from ..parallel_state import masked_compute_stream
            st = self._step_stream = masked_compute_stream(self.device, n)
        cur = torch.cuda.current_stream()
        st.wait_stream(cur)
        # the conv launchers size their tile rounds (split-K tail) for the CUs the masked stream can reach
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        # (on every library of the process: fp16-storage convs run in libp2phd_hip_f16.so, which has its own option globals)
        _ops._lib.set_option_all(b"cus", n_cu - n)
        try:
            with torch.cuda.stream(st):
                out = fn(*args, **kw)
        finally:
            _ops._lib.set_option_all(b"cus", 0)
        cur.wait_stream(st)
        return out

    def train_step(self, lr_audio, hr_audio, noise=None):
        return self._on_step_stream(self._train_step, lr_audio, hr_audio, noise)

    def _train_step(self, lr_audio, hr_audio, noise=None):
        ld = self._phase_a(lr_audio, hr_audio, noise)              # G buckets are in flight: they overlap the D backward
        self._phase_b()
        self.optimizer_D.reduce_gradients_async()
        self.optimizer_G.step(use_scaler=True)
        self.optimizer_D.step(use_scaler=True)
        if self.scaler is not None:
            self.scaler.update()                                   # once per iteration (train.py:181)
        return ld

    # ------------------------------------------------------------------------------------------
    # the same step captured once into HIP graphs and replayed: ~10^3 launches and the whole autograd walk cost no
    # host time afterwards, so the step rate no longer depends on the host core that feeds the GPU.  Graphs on one
    # memory pool -- A0: forward + first stage of the G backward, A1..: its further stages (one per gradient bucket),
    # B: D backward, Cg / Cd: the two Adam updates -- so that with data parallelism the RCCL all-reduces run BETWEEN replays,
    # outside capture: bucket i's exchange beside stage i+1, the last G bucket beside graph B, the D exchange beside Cg.
    # ------------------------------------------------------------------------------------------
    def train_step_graphed(self, lr_audio, hr_audio):
        return self._on_step_stream(self._train_step_graphed, lr_audio, hr_audio)

    def _train_step_graphed(self, lr_audio, hr_audio):
        """`train_step` through captured graphs.  Inputs are copied into static buffers; the returned loss tensors are
        static outputs of graph A0 (valid until the next call).  Mask noise is drawn inside the graph."""
        st = getattr(self, '_graph_state', None)
        key = (tuple(lr_audio.shape), tuple(hr_audio.shape), self.optimizer_G.world_size, self._bucket_plan()[0])
        if st is None or st['key'] != key:
            dev = self.device
            st = {'key': key, 'lr': torch.empty(lr_audio.shape, dtype=torch.float32, device=dev),
                  'hr': torch.empty(hr_audio.shape, dtype=torch.float32, device=dev), 'graphs': None, 'out': None, 'calls': 0}
            self._graph_state = st
        st['lr'].copy_(lr_audio, non_blocking=True)
        st['hr'].copy_(hr_audio, non_blocking=True)
        optG, optD = self.optimizer_G, self.optimizer_D
        optG.sync_hyper()
        optD.sync_hyper()
        if st['graphs'] is None:
            st['calls'] += 1
            if st['calls'] <= 2:                                   # eager steps first: workspaces, packed-weight buffers,
                return self._train_step(st['lr'], st['hr'])         # library state all exist before capture
            torch.cuda.synchronize()
            # back-to-back captures on one side stream and one pool, without the cache flush torch.cuda.graph()
            # does on entry (blocks the first capture freed must stay where its replay will write them)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            gA, ranges = [], []
            # thread-local capture mode: the RCCL watchdog thread of torch.distributed polls the events of the collectives
            # of the eager steps (hipEventQuery) while this thread captures; in the default global mode that query is
            # "not permitted when stream is capturing" and the watchdog takes the process down
            mode = "thread_local"
            with torch.cuda.stream(side):
                g0 = torch.cuda.CUDAGraph()
                g0.capture_begin(capture_error_mode=mode)
                st['out'] = self._phase_a_forward(st['lr'], st['hr'])
                stages = self._g_stages()
                for i, (run, rng) in enumerate(stages):
                    if i > 0:
                        g = torch.cuda.CUDAGraph()
                        g.capture_begin(pool=g0.pool(), capture_error_mode=mode)
                    else:
                        g = g0
                    run()
                    g.capture_end()
                    gA.append(g)
                    ranges.append(rng)
                self._loss_G = None
                gB, gCg, gCd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                gB.capture_begin(pool=g0.pool(), capture_error_mode=mode)
                self._phase_b()
                gB.capture_end()
                gCg.capture_begin(pool=g0.pool(), capture_error_mode=mode)
                optG.step_local(use_scaler=True)
                gCg.capture_end()
                gCd.capture_begin(pool=g0.pool(), capture_error_mode=mode)
                optD.step_local(use_scaler=True)
                if self.scaler is not None:
                    self.scaler.update()
                gCd.capture_end()
            torch.cuda.current_stream().wait_stream(side)
            optG.step_count -= 1                                   # capture records, it does not execute
            optD.step_count -= 1
            st['graphs'] = (gA, ranges, gB, gCg, gCd)
        gA, ranges, gB, gCg, gCd = st['graphs']
        optG.bucket_log, optD.bucket_log = [], []
        for g, (a, b) in zip(gA, ranges):
            g.replay()
            optG.reduce_range_async(a, b)                           # no-ops on one GPU
        gB.replay()
        optD.reduce_gradients_async()                               # the D exchange runs beside the generator's Adam
        optG.wait_gradients()
        gCg.replay()
        optD.wait_gradients()
        gCd.replay()
        optG.step_count += 1
        optD.step_count += 1
        # the replayed Adam changed the master weights behind the host's back: packed copies cached by ConvSpec are stale
        # for any EAGER call that follows (inference / forward / train_step between replays).  Graph A re-packs into the
        # same buffers by itself, so the replay path is unaffected by the bump.
        _ops.bump_weight_epoch()
        return st['out']

    def save(self, which_epoch):
        self.save_network(self.netG, 'G', which_epoch, self.gpu_ids)
        self.save_network(self.netD, 'D', which_epoch, self.gpu_ids)
        if getattr(self, 'scaler', None) is not None:
            # fp16 storage: scale + growth tracker of the device loss scaler, in a file of its own beside the reference's two
            # (the reference restarts GradScaler at 65536 on resume, train.py:67; a resume without the file does the same)
            torch.save(self.scaler.state_dict(), os.path.join(self.save_dir, '%s_scaler.pth' % which_epoch))

    def update_fixed_params(self):
        """pix2pixHD_model.py:521-528: after niter_fix_global epochs the global generator trains too -- a fresh Adam over
        every generator parameter (the reference resets the optimiser state as well).  Graphs captured with the old
        optimiser's buffers are dropped; data-parallel settings carry over."""
        old = self.optimizer_G
        self.optimizer_G = FlatAdam(list(self.netG.parameters()), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        self.optimizer_G.scaler, self.optimizer_G.scaler_index = getattr(self, 'scaler', None), 0
        if getattr(old, '_collectives', False) or old.world_size > 1:
            self.optimizer_G.enable_data_parallel(old.world_size, old.process_group, getattr(old, '_collectives', False) and old.world_size == 1,
                                                  wire_dtype=getattr(old, 'wire_dtype', torch.float32))
        self._graph_state = None
        if _opt(self.opt, 'verbose', False):
            print('------------ Now also finetuning global generator -----------')

    def update_learning_rate(self):
        lrd = self.opt.lr / self.opt.niter_decay
        lr = self.old_lr - lrd
        for opt in (self.optimizer_D, self.optimizer_G):
            for param_group in opt.param_groups:
                param_group['lr'] = lr
        if _opt(self.opt, 'verbose', False):
            print('update learning rate: %f -> %f' % (self.old_lr, lr))
        self.old_lr = lr

    def get_current_visuals(self):
        """Sample-0 spectrograms as numpy arrays (the reference renders them with matplotlib; that
        observability layer is out of scope, the data it plots is here)."""
        if self._visual is None:
            return {}
        lr_s, sr, hr_s, hr_pha = self._visual
        out = {'lable_spectro': 0.5 * (lr_s[0, 0] + lr_s[0, 1]).cpu().numpy(),
               'generated_spectro': 0.5 * (sr[0, 0] + sr[0, 1]).cpu().numpy()}
        if hr_s is not None:
            out['real_spectro'] = 0.5 * (hr_s[0, 0] + hr_s[0, 1]).cpu().numpy()
        sr_pha = torch.sign(sr[0, 0] - sr[0, 1])
        out['generated_pha'] = sr_pha.cpu().numpy()
        if hr_pha is not None:
            out['real_pha'] = hr_pha[0, 0].cpu().numpy()
            out['lable_pha'] = (hr_pha[0, 0] - sr_pha).cpu().numpy()
        return out


class InferenceModel(Pix2PixHDModel):
    def forward(self, inp):
        label, inst = inp
        return self.inference(label, inst)

"""Where does the near-uniform 7.8e-3 distance between the REFERENCE-style fp32 CPU gradients and the exact (fp64) gradients
of configs[1]'s generator come from?  (profiles/r03_reference_fp32_vs_fp64_gradients.txt; round-3 review, weak item 2.)

CPU only, oracle only (the restatement of the reference's networks and losses, oracle/model.py).  The generator loss is
taken apart -- G_GAN only, G_GAN_Feat only, each discriminator scale alone, feature levels alone -- and for every part
the fp32 gradient of every generator weight is compared with the fp64 gradient of the SAME part, on the input of
tests/test_gpu_fullsize.py::test_configs1_network_losses_and_both_backward_passes.  A second table separates "fp32
forward" from "fp32 backward": the fp64 backward of a loss whose forward branch decisions (ReLU masks, sign(fake - real))
are taken from the fp32 run.

    python tools/locate_fp32_gradient_distance.py > profiles/r04_fp32_vs_fp64_gradient_parts.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import mdct4 as M4  # noqa: E402
from oracle import model as OM  # noqa: E402

N = OM.N


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def parts(pG, pD, lr_s, hr_s, oo):
    """dict name -> scalar loss part, all built on ONE forward of G and D (fake attached, real detached as the reference)."""
    D = lambda x: N.multiscale_discriminator_forward(pD, x, oo.ndf, oo.n_layers_D, oo.num_D, True)
    sr = OM.netG_forward(pG, lr_s, oo)
    pred_real = D(torch.cat((lr_s, hr_s), dim=1))
    pred_fake = D(torch.cat((lr_s, sr), dim=1))
    out = {}
    out["G_GAN"] = N.gan_loss(pred_fake, True)
    out["G_GAN_Feat"] = N.feature_matching_loss(pred_fake, pred_real, oo.n_layers_D, oo.num_D, oo.lambda_feat)
    fw = 4.0 / (oo.n_layers_D + 1) * (1.0 / oo.num_D) * oo.lambda_feat
    for i in range(oo.num_D):
        out[f"G_GAN scale{i}"] = ((pred_fake[i][-1] - 1.0) ** 2).mean()
        out[f"Feat scale{i}"] = sum(fw * (pred_fake[i][j] - pred_real[i][j].detach()).abs().mean() for j in range(len(pred_fake[i]) - 1))
    for j in range(len(pred_fake[0]) - 1):
        out[f"Feat level{j} (both scales)"] = sum(fw * (pred_fake[i][j] - pred_real[i][j].detach()).abs().mean() for i in range(oo.num_D))
    # a smooth surrogate of the feature-matching term: squared instead of absolute differences (no sign() in the backward)
    out["Feat as L2 (diagnostic)"] = sum(fw * ((pred_fake[i][j] - pred_real[i][j].detach()) ** 2).mean()
                                         for i in range(oo.num_D) for j in range(len(pred_fake[i]) - 1))
    # a loss that does not pass through the discriminator at all: the generator alone
    out["G alone: mean(sr^2) (diagnostic)"] = (sr ** 2).mean()
    return out


def grads_of_parts(pG, pD, lr_s, hr_s, oo):
    pG = {k: v.detach().clone().requires_grad_(True) for k, v in pG.items()}
    P = parts(pG, pD, lr_s, hr_s, oo)
    keys = [k for k in pG if k.endswith("weight")]
    out = {}
    for name, loss in P.items():
        g = torch.autograd.grad(loss, [pG[k] for k in keys], retain_graph=True)
        out[name] = dict(zip(keys, [t.detach() for t in g]))
    return out, {k: float(v) for k, v in P.items()}


class BranchTape:
    """Records the (Leaky)ReLU branch of every element during one run of the oracle networks and replays it in another:
    `replay` runs take the branch decisions of the recorded run, whatever the sign of their own pre-activation."""

    def __init__(self):
        self.masks, self.mode, self.i = [], None, 0
        self._relu, self._lrelu = N.F.relu, N.F.leaky_relu

    def relu(self, x, *a, **k):
        return self._apply(x, 0.0)

    def lrelu(self, x, slope=0.01, *a, **k):
        return self._apply(x, slope)

    def _apply(self, x, slope):
        if self.mode == "record":
            m = x > 0
            self.masks.append(m)
        else:
            m = self.masks[self.i]
            self.i += 1
        return x * torch.where(m, torch.ones((), dtype=x.dtype), torch.full((), slope, dtype=x.dtype))

    def __enter__(self):
        N.F.relu, N.F.leaky_relu = self.relu, self.lrelu
        return self

    def __exit__(self, *exc):
        N.F.relu, N.F.leaky_relu = self._relu, self._lrelu
        return False


def main():
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    oo = OM.default_opt(ngf=48, netG="global", n_downsample_global=4, n_blocks_global=9, mask=False)
    pG = N.init_params(OM.netG_spec(oo), seed=1)
    pD = N.init_params(OM.netD_spec(oo), seed=2)
    hr, lr, _ = OM.synthetic_batch(1, oo, seed=5)
    w = M4.kbdwin(oo.win_length)
    hr_s, _, _ = OM.to_spectro(hr, oo, w, mask=False)
    lr_s, _, _ = OM.to_spectro(lr, oo, w, mask=False)
    t0 = time.time()
    g32, l32 = grads_of_parts(pG, pD, lr_s, hr_s, oo)
    t1 = time.time()
    d = lambda t: {k: v.double() for k, v in t.items()}
    g64, l64 = grads_of_parts(d(pG), d(pD), lr_s.double(), hr_s.double(), oo)
    t2 = time.time()
    print(f"# configs[1] generator (102 627 170 parameters) + 2-scale D, 512x256, B=1, torch {torch.__version__} CPU, "
          f"{torch.get_num_threads()} threads; fp32 pass {t1 - t0:.1f} s, fp64 pass {t2 - t1:.1f} s")
    print("# relative L2 distance of the fp32 gradient from the fp64 gradient OF THE SAME LOSS PART, per generator weight")
    names = list(g32)
    keys = list(g32[names[0]])
    show = [keys[0], keys[1], keys[4], keys[5], keys[13], keys[22], keys[23], keys[25], keys[27]]
    print(f"{'loss part':36s} {'value fp32':>12s} {'value fp64':>12s} | " + " ".join(f"{k.replace('model.', 'm').replace('.conv_block', '.cb').replace('.weight', ''):>9s}" for k in show)
          + " |   min..max over all 28 weights")
    for n in names:
        e = {k: rel(g32[n][k], g64[n][k]) for k in keys}
        print(f"{n:36s} {l32[n]:12.6f} {l64[n]:12.6f} | " + " ".join(f"{e[k]:9.2e}" for k in show) + f" | {min(e.values()):.2e} .. {max(e.values()):.2e}")
    # ---- arithmetic or branch decisions?  fp64 arithmetic with the (Leaky)ReLU branches of the fp32 forward ------------------
    tape = BranchTape()
    sel = ["G alone: mean(sr^2) (diagnostic)", "G_GAN", "Feat as L2 (diagnostic)"]
    with tape:
        tape.mode = "record"
        g32m, _ = grads_of_parts(pG, pD, lr_s, hr_s, oo)                  # same numbers as g32 (x * mask == relu(x))
        tape.mode, tape.i = "replay", 0
        g64m, _ = grads_of_parts(d(pG), d(pD), lr_s.double(), hr_s.double(), oo)
    print("# the same distances with the branch of every ReLU / LeakyReLU element taken from the fp32 forward in BOTH runs")
    print("# (fp64 arithmetic, fp32 decisions): what is left is arithmetic; what disappeared was a flipped branch")
    for n in sel:
        e = {k: rel(g32m[n][k], g64m[n][k]) for k in keys}
        print(f"{n:36s} {'':12s} {'':12s} | " + " ".join(f"{e[k]:9.2e}" for k in show) + f" | {min(e.values()):.2e} .. {max(e.values()):.2e}")
    flips = []
    with tape:                                                              # count the flipped branches per layer
        tape.mode, tape.i = "count", 0
        def counting(x, slope):
            m = tape.masks[tape.i]; tape.i += 1
            own = x > 0
            flips.append((int((own != m).sum()), m.numel(), tuple(x.shape)))
            return x * torch.where(own, torch.ones((), dtype=x.dtype), torch.full((), slope, dtype=x.dtype))
        tape._apply = counting
        with torch.no_grad():
            parts(d(pG), d(pD), lr_s.double(), hr_s.double(), oo)
    n_g = 1 + 2 * oo.n_downsample_global + oo.n_blocks_global          # ReLUs of the generator: c7, down, one per block, up
    print("# elements whose fp32 branch differs from the fp64 branch, per activation layer in execution order (G: 18 ReLU -- c7, 4 down, 9 blocks, 4 up --, then the LeakyReLUs of D(real), first 9 shown)")
    for i, (nf, tot_el, shp) in enumerate(flips[:n_g + 9]):
        name = f"G relu {i:2d}" if i < n_g else f"D(real) lrelu {i - n_g:2d}"
        print(f"  {name:18s} {str(shp):24s} {nf:8d} of {tot_el:10d} = {nf / tot_el:.2e}   sqrt(share) {(nf / tot_el) ** 0.5:.2e}")
    # how much of the whole generator gradient each part is (fp64 norms), at the input layer and at the output layer
    tot = {k: g64["G_GAN"][k] + g64["G_GAN_Feat"][k] for k in keys}
    print("# share of each part in the norm of the whole generator gradient (fp64): first layer / last layer")
    for n in names:
        print(f"{n:36s} {float(g64[n][keys[0]].norm() / tot[keys[0]].norm()):8.3f} {float(g64[n][keys[-1]].norm() / tot[keys[-1]].norm()):8.3f}")
    e_tot = {k: rel(g32["G_GAN"][k] + g32["G_GAN_Feat"][k], tot[k]) for k in keys}
    print("# whole generator loss (G_GAN + G_GAN_Feat), every weight:")
    for k in keys:
        print(f"  {k:34s} {e_tot[k]:.2e}")


if __name__ == "__main__":
    main()

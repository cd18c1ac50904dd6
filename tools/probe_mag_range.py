"""Probe (diagnostic): how AA_OUT_MAG = v_sqrt_f32(re^2 + im^2) behaves outside the audio range,
against hypot semantics (the reference's c.norm(), stft.rs:314-318).  Run on a GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    import torch
    import audio_analyzer_rs_amd as aa

    rng = np.random.default_rng(0)
    for n in (1024, 2048, 256, 64):
        base = rng.uniform(-1, 1, n * 8).astype(np.float32)
        plan = aa.Stft(n, n, window="rect", output="mag")
        cplan = aa.Stft(n, n, window="rect", output="complex")
        for k in (-45, -40, -38, -30, -25, -22, -21, -20, -19, -18, -10, 0, 10, 15, 16, 17, 18, 19, 20, 30):
            x = (base.astype(np.float64) * 10.0 ** k).astype(np.float32)
            got = plan(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
            cg = cplan(torch.from_numpy(x).cuda()).cpu().numpy()
            hyp = np.hypot(cg.real.astype(np.float32), cg.imag.astype(np.float32)).astype(np.float64)
            ref = np.abs(np.fft.rfft(x.astype(np.float64).reshape(-1, n), axis=-1))
            with np.errstate(all="ignore"):
                rel = np.abs(got - hyp) / np.where(hyp > 0, hyp, 1)
            fin = np.isfinite(got)
            print(f"n={n:5d} scale 1e{k:+03d}: max|X| {ref.max():.3e} min|X| {ref.min():.3e}  finite {fin.mean():.3f} "
                  f"zeros {np.mean(got == 0):.3f} (hypot zeros {np.mean(hyp == 0):.3f})  max rel vs hypotf(own complex) "
                  f"{np.nanmax(np.where(fin, rel, np.nan)) if fin.any() else float('nan'):.3e}  hypot finite {np.isfinite(hyp).mean():.3f}")
        # non-finite inputs
        for name, val in (("nan", np.nan), ("+inf", np.inf), ("-inf", -np.inf)):
            x = base.copy()
            x[5] = val                                   # frame 0 only
            got = plan(torch.from_numpy(x).cuda()).cpu().numpy()
            cg = cplan(torch.from_numpy(x).cuda()).cpu().numpy()
            hyp = np.hypot(cg.real, cg.imag)
            print(f"n={n:5d} {name} in frame 0: frame0 finite {np.isfinite(got[0]).mean():.3f} nan {np.isnan(got[0]).mean():.3f} "
                  f"inf {np.isinf(got[0]).mean():.3f} | hypot of own complex: nan {np.isnan(hyp[0]).mean():.3f} inf {np.isinf(hyp[0]).mean():.3f} "
                  f"| other frames finite {np.isfinite(got[1:]).all()}")
        # denormal inputs
        x = (base.astype(np.float64) * 1e-41).astype(np.float32)
        got = plan(torch.from_numpy(x).cuda()).cpu().numpy()
        print(f"n={n:5d} denormal input (|x| <= 1e-41): output zeros {np.mean(got == 0):.3f}, max {got.max():.3e}; "
              f"f64 max|X| {np.abs(np.fft.rfft(x.astype(np.float64).reshape(-1, n), axis=-1)).max():.3e}")


if __name__ == "__main__":
    main()

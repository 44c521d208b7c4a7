"""The float64 matrix instruction holds the vector ALU: time the dense strip product Kuf_bar = G A (gemm_strip_kernel<3>,
alone on the device) in builds that add N dummy integer vector adds per K-tile (tools/build_variant.sh valuN gemm_strip.hip
-DGS_EXTRA_VALU=N).  A K-tile is 128 v_mfma_f64_16x16x4_f64 per wavefront = 8192 matrix cycles; if vector instructions
issued beside them were free (as they are next to f32 / bf16 MFMAs) the time would not move.
    GPITCH_AMD_LIB=tools/ab/lib_valu48.so python tools/valu_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    model = pdgp_from_problem(make_problem(32768, 512, 12, num_partials=20, seed=0))
    model.za.fixed = True; model.zc.fixed = True
    model._pack()
    h = model._handle
    h.check(h.lib.gp_pdgp_set_overlap(model._plan, 0))
    for _ in range(3):
        model._elbo(True, sync=False)
    torch.cuda.synchronize()
    h.check(h.lib.gp_timers_enable(h.h, 1)); h.check(h.lib.gp_timers_reset(h.h))
    for _ in range(10):
        model._elbo(True, sync=False)
    torch.cuda.synchronize()
    h.check(h.lib.gp_timers_enable(h.h, 0))
    t = h.timers()
    print("lib=%s" % os.environ.get("GPITCH_AMD_LIB", "product"),
          "(ms per step, kernels alone)", " ".join("%s=%.3f" % (k, t[k][0] / 10.0) for k in ("kuf_bar", "cond_A", "cond_LTA", "nt_gemm")))


if __name__ == "__main__":
    main()

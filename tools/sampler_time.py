"""configs[3]'s sampler alone (for rocprofv3): python3 tools/sampler_time.py [samples]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from glenet_amd import dense_path as dp, synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    if len(sys.argv) > 2 and sys.argv[2] == "modules":        # the decoder module by module behind the extractors' kernels
        dp.CVAE.FUSED_SAMPLER = False
    if os.environ.get("GLX_POINTNET_FORM"):                   # 0: W3 through the LDS ring, 1: W3 in registers
        from glenet_amd import _lib
        _lib.load().glx_pointnet_feat_set_form(int(os.environ["GLX_POINTNET_FORM"]))
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    pts = torch.from_numpy(synth.cvae_objects(4096, 2000, 512, with_labels=True)[0]).to(dev)
    model = dp.CVAE(4, 8).to(dev).eval()
    eps = torch.randn((n, 4096, 8), device=dev)
    with torch.no_grad():
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for s_ in range(n):
                model.sample(pts, eps[s_])
            t1.record(); torch.cuda.synchronize()
            print("ms per sample %.3f" % (t0.elapsed_time(t1) / n))

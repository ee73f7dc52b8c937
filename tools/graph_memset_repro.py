"""Does a hipMemsetAsync recorded into a HIP graph zero all of its bytes when the graph is launched?  torch only + ctypes on
the HIP runtime torch has loaded.  Background: tools/graph_reduce_repro.py (reductions that zero their semaphores with a
memset in front of the kernel lose 3/4 of their outputs from the second replay on)."""
import ctypes

import torch

dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
print(torch.__version__, torch.version.hip, torch.cuda.get_device_name(0))
for nbytes in (4, 16, 64, 256, 4096, 1 << 20):
    for kind in ("hipMemsetAsync", "hipMemsetD32Async"):
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(dev)

        def op():
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if kind == "hipMemsetAsync":
                rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, nbytes, st)
            else:
                rc = hip.hipMemsetD32Async(ctypes.c_void_p(buf.data_ptr()), 0, nbytes // 4, st)
            assert rc == 0, rc
        buf.fill_(255)
        op()
        torch.cuda.synchronize()
        eager_left = int((buf != 0).sum())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            op()
        left = []
        for rep in range(3):
            buf.fill_(255)
            g.replay()
            torch.cuda.synchronize()
            left.append(int((buf != 0).sum()))
        print("%-18s %8d bytes: non-zero bytes left -- eager %d, graph launches %s" % (kind, nbytes, eager_left, left), flush=True)
        if nbytes == 64:
            for fillv in (255, 1, 170):
                buf.fill_(fillv)
                g.replay()
                torch.cuda.synchronize()
                print("      buffer pre-filled with %3d, after the graph launch: %s" % (fillv, buf.cpu().tolist()), flush=True)

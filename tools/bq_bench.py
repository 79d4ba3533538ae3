import sys, os, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, "minimal-sdr_amd/python"); sys.path.insert(0, ".")
import msdr, bench
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = msdr.Context(0, stream.cuda_stream)
bq = np.array(bench.reference_biquads(msdr), np.float32)
for ch, n in ((4096, 1 << 18), (1, 1 << 28)):
    x = torch.rand((ch, n), device=dev) * 2 - 1
    y = torch.empty_like(x)
    f = msdr.BiquadDf1F32(ctx, bq, ch)
    fn = ctx.lib.msdr_biquad_df1_f32_process
    def step():
        assert fn(f.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_uint32(n)) == 0, ctx.lib.msdr_last_error()
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("biquad_df1 2 stages: %d ch x %d: %.3f ms, %.1f Gsamples/s, %.1f GB/s (%.1f %% of 8 TB/s)" % (ch, n, ms, ch * n / ms / 1e6, 8 * ch * n / ms / 1e6, 8 * ch * n / ms / 1e6 / 80))

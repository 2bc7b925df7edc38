#!/usr/bin/env python3
"""Micro-benchmark of the projection core (sepr_linear_fwd: plain prologue, bias epilogue) on the
separator's shapes.  Prints TFLOP/s per shape; pick the library build with SEPR_LIB_VARIANT."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from sepreformer_amd import lib as L  # noqa: E402

SHAPES = [(512000, 768, 128), (512000, 128, 384), (512000, 128, 128), (512000, 384, 128), (512000, 256, 128),
          (256000, 1024, 128), (64000, 768, 128), (16000, 768, 128), (8192, 8192, 4096)]


if os.environ.get("GEMM_SHAPES"):      # "M,N,K;M,N,K;..."
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["GEMM_SHAPES"].split(";")]
SKIP_F32 = os.environ.get("GEMM_SKIP_F32", "0") == "1"


def main():
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    print("variant", os.environ.get("SEPR_LIB_VARIANT", "default"), L.build_info())
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda")
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = float("nan")
        if not SKIP_F32:
            for _ in range(3):
                L.check(lib.sepr_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, st), "lin")
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                lib.sepr_linear_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * M * N * K / ms / 1e9
        gbs = (M * K + M * N) * 4 / ms / 1e6
        line = f"  M={M:7d} N={N:5d} K={K:5d}  f32 {ms:8.3f} ms {tf:7.1f} TF {gbs:6.0f} GB/s"
        if K % 64 == 0:
            from sepreformer_amd.pack import pack_x3
            wp = pack_x3(w)
            for _ in range(3):
                L.check(lib.sepr_linear_x3_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, st), "x3")
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                lib.sepr_linear_x3_fwd(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), M, N, K, st)
            e1.record()
            torch.cuda.synchronize()
            ms3 = e0.elapsed_time(e1) / reps
            line += f" | bf16x3 {ms3:8.3f} ms {2.0 * M * N * K / ms3 / 1e9:7.1f} TF-eq {(M * K + M * N) * 4 / ms3 / 1e6:6.0f} GB/s"
        print(line)
        del x, w, b, y


if __name__ == "__main__":
    main()

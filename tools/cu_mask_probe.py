#!/usr/bin/env python3
"""Do CU-masked streams (ccd_amd/streams.py) restrict kernels to their compute units on this box, and what do two row-owner
kernels cost side by side on disjoint halves of the chip?   python tools/cu_mask_probe.py [cu:16 | xcd:4]   -> JSON lines"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ccd_amd import ops, streams


def timed(fn, reps=20, streams_=()):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "cu:16"
    layout, _, n = spec.partition(":")
    dev = torch.device("cuda:0")
    R, E, H = 131072, 384, 1536
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda *s, dt=torch.bfloat16, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dt).to(dev)
    y, w1, w2 = mk(R, E), mk(H, E, sc=0.05), mk(E, H, sc=0.05)
    b1, b2, ga, be = (mk(n_, dt=torch.float32) for n_ in (H, E, E, E))
    resid = mk(R, E, dt=torch.float32)

    def mlp(store_u=False):
        return ops.mlp_fused(y, w1, b1, w2, b2, resid=resid, rowscale=None, rows_per_sample=256, gamma=ga, beta=be, eps=1e-6,
                             store_u=store_u)

    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB

    def stream_op():
        big.mul_(1.0001)

    out = {"spec": spec, "mlp_full_chip_ms": round(timed(mlp), 4), "stream_full_chip_ms": round(timed(stream_op), 4)}
    s_stream, t_stream, s_cus, t_cus = streams.partition(dev, teacher_per_xcd=int(n), layout=layout)
    out["cus"] = [s_cus, t_cus]

    def on(stream, reserve, fn):
        def run():
            with torch.cuda.stream(stream), ops.policy(cu_reserve=reserve, cu_reserve_window=-1):
                fn()
        return run

    out["mlp_on_student_part_ms"] = round(timed(on(s_stream, t_cus, mlp)), 4)
    out["mlp_on_teacher_part_ms"] = round(timed(on(t_stream, s_cus, mlp)), 4)
    out["stream_on_teacher_part_ms"] = round(timed(on(t_stream, s_cus, stream_op)), 4)
    a, b = on(s_stream, t_cus, mlp), on(t_stream, s_cus, mlp)
    out["mlp_both_parts_ms"] = round(timed(lambda: (a(), b())), 4)
    c = on(t_stream, s_cus, stream_op)
    out["mlp_student_part_with_stream_on_teacher_part_ms"] = round(timed(lambda: (a(), c())), 4)
    # unmasked second stream for comparison (what round 3 measured: each kernel as much slower as the other takes)
    side = torch.cuda.Stream(device=dev)
    d = on(side, 0, mlp)
    out["mlp_two_unmasked_streams_ms"] = round(timed(lambda: (mlp(), d())), 4)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

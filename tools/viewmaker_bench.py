"""Steady-state cost of the f3 view maker (row f3 of SURVEY.md §8): one call = parameters of the NEXT batch drawn on the host, weather
layers computed by the worker pool, the two device kernels of THIS batch.  Prints one JSON line per mode.

    python tools/viewmaker_bench.py [--batch 256] [--iters 12] [--gap-ms 49]

`--gap-ms`: what the training step between two calls costs (the pool works during it); 0 = back to back (the pool's own throughput).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--gap-ms", type=float, default=49.0)
    ap.add_argument("--workers", type=int, default=None)
    args = ap.parse_args()
    from ccd_amd.dataset import DeviceViewMaker
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(0)
    images = torch.from_numpy(rs.randint(0, 256, (args.batch, 32, 128, 3)).astype(np.uint8))
    masks = torch.from_numpy((rs.rand(args.batch, 32, 128) > 0.5).astype(np.float32))
    for gap in (args.gap_ms, 0.0):
        maker = DeviceViewMaker(32, 128, severity=5, seed=1, device=dev, workers=args.workers)
        calls, layers = [], []
        for it in range(args.iters + 2):
            if gap > 0:
                time.sleep(gap * 1e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = maker(images, masks)
            torch.cuda.synchronize()
            if it >= 2:                                       # (the first call forks the pool and draws two batches)
                calls.append((time.perf_counter() - t0) * 1e3)
        workers = maker._farm.workers if maker._farm is not None else 0
        maker._farm.close()
        print(json.dumps({"what": "DeviceViewMaker call, steady state", "batch": args.batch, "gap_ms_between_calls": gap, "workers": workers,
                          "cores": os.cpu_count(), "call_ms_median": round(float(np.median(calls)), 2), "call_ms_max": round(float(np.max(calls)), 2),
                          "images_per_s_at_median": round(args.batch / (float(np.median(calls)) + gap) * 1e3, 1),
                          "views_shape": list(out[0].shape)}), flush=True)


if __name__ == "__main__":
    main()

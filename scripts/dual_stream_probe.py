"""Probe: does running the B=128 chunk as TWO concurrent B=64 graph replays (two HIP streams) beat one B=128 replay?  The small row /
attention kernels of one half would overlap the GEMMs of the other.  Pessimistic setup: the halves use separate weight copies."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, den = bench.build_model(dev)
    sig = M.get_sigmas_exponential(10, 1e-3, 80.0).to(dev)
    img, goal, x0 = bench.synthetic_inputs(dev, 128)

    def run(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    one = run(lambda: M.sample_ddim(den, {"state_images": img}, x0, goal, sig, disable=True))
    print(f"one B=128 graph: {one:.2f} ms/chunk = {1280 / one:.1f} denoise-steps/s x1000")
    for parts in (2, 4):
        b = 128 // parts
        dens = [den] + [bench.build_model(dev)[1] for _ in range(parts - 1)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        ins = [(img[i * b:(i + 1) * b].contiguous(), goal[i * b:(i + 1) * b].contiguous(), x0[i * b:(i + 1) * b].contiguous()) for i in range(parts)]
        # capture each graph on its own
        for d, (im, go, xx) in zip(dens, ins):
            M.sample_ddim(d, {"state_images": im}, xx, go, sig, disable=True)
        torch.cuda.synchronize()

        def both():
            cur = torch.cuda.current_stream()
            for s, d, (im, go, xx) in zip(streams, dens, ins):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    M.sample_ddim(d, {"state_images": im}, xx, go, sig, disable=True)
            for s in streams:
                cur.wait_stream(s)

        def serial():
            for d, (im, go, xx) in zip(dens, ins):
                M.sample_ddim(d, {"state_images": im}, xx, go, sig, disable=True)
        ts = run(serial); tb = run(both)
        print(f"{parts} x B={b}: serial {ts:.2f} ms, concurrent {tb:.2f} ms = {1280 / tb:.1f} denoise-steps/s x1000")


if __name__ == "__main__":
    main()

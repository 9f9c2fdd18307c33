#!/usr/bin/env python3
"""A few hundred full train() iterations of the headline configuration on synthetic data: losses stay finite, epsilon grows, and
device memory does not (leak check for the per-step buffers: engine arena, ghost stashes, sampler draws)."""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
EXTRA = sys.argv[2].split() if len(sys.argv) > 2 else []        # e.g. "--compute_dtype bf16 --storage_dtype bf16 --im_size 128"
with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0, extra=EXTRA)
B = img.shape[0]
lab = torch.zeros(B, dtype=torch.long)
mem = []
t0 = time.perf_counter()
for it in range(N):
    tr.train(0, it, img, lab, use_dp=True)
    if it % 50 == 49:
        torch.cuda.synchronize()
        tr.flush_stats()
        st = tr.logger.stats
        mem.append(torch.cuda.memory_allocated() / 2 ** 20)
        print("it %4d  %.1f it/s  alloc %.0f MiB (peak %.0f)  D adv %.4f  penalty %.4f  eps %.3f" % (
            it + 1, (it + 1) / (time.perf_counter() - t0), mem[-1], torch.cuda.max_memory_allocated() / 2 ** 20,
            float(st.get("D Adv Loss", 0.0)), float(st.get("D Penalty", 0.0)), tr.privacy_engine.get_privacy_spent(opt.delta)[0]))
        tr.reset_stats()
ok = all(torch.isfinite(p).all().item() for p in list(tr.D.parameters()) + list(tr.G.parameters()))
print("finite weights:", ok, " memory growth over the run: %.1f MiB" % (mem[-1] - mem[0]))
assert ok and mem[-1] - mem[0] < 64

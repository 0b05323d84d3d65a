"""Per-CU timeline of the A-direct implicit-GEMM kernel from the s_memtime stamps of a -DLIP_DBG2 build
(csrc/lip_mfma.hip: kernel entry, K loop start, K loop end, epilogue issued, [stores drained]; HW_ID; XCC_ID):
phase durations per block, how many of a SIMD's resident waves are inside the K loop, K-tile period per wave."""
import sys
import numpy as np

f = sys.argv[1]
ktiles = int(sys.argv[2]) if len(sys.argv) > 2 else 18
d = np.fromfile(f, dtype=np.uint64).reshape(-1, 4, 6)
t0, t1, t2, t3 = [d[:, :, i].astype(np.int64) for i in range(4)]
hw = d[:, :, 4].astype(np.int64)
xcc = d[:, :, 5].astype(np.int64) & 0xf
issue = d[:, :, 5].astype(np.int64) >> 8
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; se = (hw >> 13) & 7
print(f"{f}: {d.shape[0]} blocks")
for name, x in (("prologue (entry -> K loop)", t1 - t0), ("K loop", t2 - t1), ("setup (entry -> first load issued)", issue), ("epilogue total", t3 - t2), ("life", t3 - t0)):
    x = x[:, 0]
    print(f"  {name:28s} mean {x.mean():8.0f}  median {np.median(x):8.0f}  p10 {np.percentile(x, 10):8.0f}  p90 {np.percentile(x, 90):8.0f} cycles")
print(f"  K-tile period per wave: {(t2 - t1)[:, 0].mean() / ktiles:.0f} cycles ({8 * 64} of them MFMA)")
key = (xcc * 1000000 + se * 10000 + cu * 10)[:, 0]
ks, cnt = np.unique(key, return_counts=True)
print(f"  {len(ks)} CUs, {cnt.min()}..{cnt.max()} blocks each")
inloop = {}; resident = {}
for k in ks[:: max(1, len(ks) // 16)]:
    sel = np.where(key == k)[0]
    for s_ in range(4):
        m = simd[sel] == s_
        for a, b, acc in ((t1[sel][m], t2[sel][m], inloop), (t0[sel][m], t3[sel][m], resident)):
            ev = sorted([(x, 1) for x in a] + [(x, -1) for x in b])
            cur = 0; last = ev[0][0]
            for t, dl in ev:
                acc[cur] = acc.get(cur, 0) + (t - last); last = t; cur += dl
for name, acc in (("in the K loop", inloop), ("resident", resident)):
    T = sum(acc.values())
    print(f"  fraction of SIMD time with k waves {name}: " + ", ".join(f"{k}: {v / T:.3f}" for k, v in sorted(acc.items()) if v / T >= 0.002),
          f"(mean {sum(k * v for k, v in acc.items()) / T:.2f})")
span = t3.max(axis=None) - t0.min(axis=None)

import numpy as np, json
a = np.load('/tmp/p_ks.npy'); b = np.load('/tmp/p_noks.npy'); m = json.load(open('/tmp/p_ks.npy.json'))
d = np.abs(a - b)
print("prim floats", a.size, "max abs diff", d.max(), "count > 1e-3:", int((d > 1e-3).sum()), "count > 0.5:", int((d > 0.5).sum()))
big = np.where(d > 1e-3)[0]
print("first offsets", big[:20])
for name in ("dphi_off", "amax_off"):
    offs = sorted((v, k) for k, v in (m.get(name) or {}).items())
    for i in big[:20]:
        prev = [(o, k) for o, k in offs if o <= i]
        if prev: print(name, "offset", i, "in/after tensor", prev[-1], "a", a[i], "b", b[i])

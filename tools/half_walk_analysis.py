import numpy as np
f = np.load("gpurun_out/walk_prof_uniform_1000000_v4096.npy").astype(float)*0.01
lo = np.load("gpurun_out/walk_prof_uniform_1000000_v16781312.npy").astype(float)*0.01
hi = np.load("gpurun_out/walk_prof_uniform_1000000_v33558528.npy").astype(float)*0.01
t, a, b = f[:,1], lo[:,1], hi[:,1]
print("mean full %.2f lo %.2f hi %.2f" % (t.mean(), a.mean(), b.mean()))
o = np.argsort(-t)
for k in (50, 200, 1000, 3000):
    i = o[:k]
    print("top %d: full %.1f  lo %.1f hi %.1f  max(lo,hi) %.1f sum %.1f" % (k, t[i].mean(), a[i].mean(), b[i].mean(), np.maximum(a[i], b[i]).mean(), (a[i]+b[i]).mean()))
print("all: max(lo,hi)/full median", np.median(np.maximum(a,b)/t), " sum/full median", np.median((a+b)/t))

import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
ctx = hip.Context(); cq = hip.CommandQueue(ctx)
for nb in (64, 1000, 16384):
    buf = hip.Buffer(ctx, nb * 4)
    for rep in range(3):
        call.col_debug_xcc_census(cq.stream, buf.ptr, nb)
        x = hip.read_buffer(cq, buf, np.uint32, nb)
        agree = [(x[i::8] == x[i]).mean() for i in range(8)]
        print(nb, "first 16:", x[:16].tolist(), "share of blocks b%8==i on the XCD of block i:", np.round(agree, 3).tolist(),
              "blocks per XCD:", np.bincount(x, minlength=8).tolist())

import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from collision_amd import hip
from collision_amd._lib import call
from collision_amd.collision import Collider
import bench
ctx=hip.Context(); cq=hip.CommandQueue(ctx)
for n in (1000000,):
    coords,radii=bench.uniform_scene(n)
    cb,rb=hip.Buffer(ctx,hostbuf=coords),hip.Buffer(ctx,hostbuf=radii)
    nb,pb=hip.Buffer(ctx,4),hip.Buffer(ctx,(1<<17)*8)
    col=Collider(ctx,n,64,256)
    col.get_collisions(cq,cb,rb,nb,pb,1<<17); cq.finish()
    for mode in (0,1,2,3):
        stats=hip.Buffer(ctx,hostbuf=np.zeros(8,np.uint64))
        z=np.zeros(1,np.uint32)
        def run():
            call.col_fill(cq.stream, nb.ptr, z.ctypes.data, 4, 1)
            call.col_traverse_stats(cq.stream,pb.ptr,nb.ptr,1<<17,col._bounds_buf.ptr,n,4,stats.ptr,mode)
        run(); cq.finish()
        ms=bench.time_events(hip,cq,run,10)
        s=hip.read_buffer(cq,stats,np.uint64,8)
        np_=(n+63)//64*11.0; print('mode',mode,'ms',round(ms,4),'per packet: steps',round(s[0]/np_,1),'descents',round(s[1]/np_,1),'leaf tests',round(s[2]/np_,1),'leaf hits',round(s[3]/np_,2),'within 1k/2k/4k/8k of the block start:', [round(float(x)/max(float(s[0]),1),3) for x in s[4:8]], 'pairs', hip.read_buffer(cq,nb,np.uint32,1))

import sys, numpy as np, torch
sys.path.insert(0,'.')
import __graft_entry__ as g
pkg=g.load_package()
svc=pkg.HipCompressionService(4,0)
lib,h=pkg.lib(),svc.ctx.handle
for name,fill,seed,n,bb in [("text256",lib.dczu_fill_text,0xD0C2,1<<30,4<<20),("text1024x1M",lib.dczu_fill_text,0xD0C2,1<<30,1<<20),("rand256",lib.dczu_fill_java_random,42,256<<20,1<<20),("low256",lib.dczu_fill_lowentropy,0xD0C5,1<<30,4<<20)]:
    t=torch.empty(n,dtype=torch.uint8,device="cuda"); fill(h,t.data_ptr(),n,seed,0,None)
    blk=svc.compress_device(t,bb); K=blk.num_chunks
    orig=torch.full((K,),bb,dtype=torch.int32,device="cuda")
    out,st,ep=svc.decompress_device(blk.payload,blk.comp_off,blk.comp_size,orig,blk.code_lengths,bb)
    torch.cuda.synchronize()
    e=ep.cpu().numpy()[:K].astype(np.uint64)
    win=e & 0xFFFFFF; rounds=(e>>24)&0xFFFFFF; mx=e>>48
    print(name,"K",K,"windows/block %.1f rounds/window %.2f max rounds in a window %d"%(win.mean(), rounds.sum()/win.sum(), mx.max()), "ok", bool(torch.equal(out[:n],t)))

import sys, numpy as np
sys.path.insert(0,'.')
import __graft_entry__ as g
import torch
pkg=g.load_package(); orc=g.load_oracle()
svc=pkg.HipCompressionService(1,0)
def run(name,data,bb):
    t=torch.from_numpy(data).cuda()
    blk=svc.compress_device(t,bb); torch.cuda.synchronize()
    tot=int(blk.total.item())
    pay=blk.payload[:tot].cpu().numpy()
    op,os_,oo,ol=orc.compress_blocks(data,bb)
    ok = pay.size==op.size and (pay==op).all()
    msg=""
    if not ok and pay.size==op.size:
        bad=np.nonzero(pay!=op)[0]
        msg="nbad=%d first=%d last=%d  hip=%s orc=%s"%(bad.size,bad[0],bad[-1],pay[bad[0]:bad[0]+8].tobytes().hex(),op[bad[0]:bad[0]+8].tobytes().hex())
    print(name, data.size, bb, "maxlen",ol.max(), "OK" if ok else "FAIL", pay.size, op.size, msg)
    return blk
run("abcd",np.frombuffer(b"AAAABBBBCCCCDDDD",dtype=np.uint8),16)
run("hello",np.frombuffer(("Hello World! "*100).encode(),dtype=np.uint8),1300)
for n in [1,16,64,100,1024,1025,4096,32768,32769,65536,100000]:
    run("text",orc.gen_text(1,0,n),n)
    run("rand",orc.java_random_bytes(42,n),n)
    run("low",orc.gen_lowentropy(2,0,n),n)
run("text-mb",orc.gen_text(0xD0C2,0,3*65536+1234),65536)

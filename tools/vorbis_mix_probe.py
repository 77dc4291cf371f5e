import sys; sys.path.insert(0,'.')
import numpy as np, torch, symphonia_amd as sa
ctx=sa.Context(0); ctx.use_torch_stream()
def run(p_ll, p_ss, label):
    nch, nb = 64, 4096
    rng=np.random.default_rng(1)
    flags=np.zeros((nch,nb),np.uint8); cur=np.ones(nch,bool)
    for b in range(nb):
        r=rng.random(nch); cur=np.where(cur, r<p_ll, r>=p_ss); flags[:,b]=cur
    v=sa.VorbisDsp(ctx,8,11)
    so,po=v.layout(flags,np.full(nch,-1))
    ss,ps=int(so[:,-1].max()),int(po[:,-1].max())
    spectra=torch.randn((nch,ss),device='cuda')*0.1
    dfl=torch.from_numpy(flags).cuda(); prev=torch.full((nch,),-1,dtype=torch.int32,device='cuda')
    ov=torch.zeros((nch,1024),device='cuda'); pcm=torch.zeros((nch,ps),device='cuda')
    def step():
        prev.fill_(-1); v.synth(spectra,dfl,prev,ov,ps,pcm)
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/20
    byts=4*(so[:,-1].sum()+po[:,-1].sum())
    print(label,'long frac %.2f'%flags.mean(),'ms %.3f'%t,'TB/s %.2f'%(byts/t/1e9), 'blocks', nch*nb)
run(1.0,0.0,'all long   ')
run(0.9,0.7,'config 4   ')
run(0.0,1.0,'all short  ')
run(0.5,0.5,'half/half  ')

import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from conftest import Golden
from oracle import explainn_oracle as orc
from explainn_amd import ExplaiNN, get_optimizer
g=Golden("small_u8_k19_mse")
m=ExplaiNN(g.U,g.k,g.L,g.T); m.load_state_dict({k:torch.from_numpy(np.array(v)) for k,v in g.sd().items()}); m=m.cuda(); m.dropout_p=0.0
opt=get_optimizer(m.parameters(),0.003); crit=torch.nn.MSELoss()
sd={k:v.copy() for k,v in g.sd().items()}; st=orc.adam_init(sd)
for step in range(1,21):
    i=(step-1)%g.n_batches
    x=torch.from_numpy(g.onehot(i)).cuda(); y=torch.from_numpy(g.targets(i).astype(np.float32)).cuda()
    m.train(); pred=m(x); loss=crit(pred,y); opt.zero_grad(); loss.backward()
    # oracle gradients at the SAME parameters as the GPU model (not its own trajectory)
    sdg={k:v.detach().cpu().numpy() for k,v in m.state_dict().items()}
    lg,cache,nb=orc.forward(sdg,g.onehot(i),training=True,return_cache=True)
    _,dl=orc.mse(lg,g.targets(i)); gr=orc.backward(cache,dl)
    errs={k:float(np.abs(p.grad.cpu().numpy().reshape(-1)-gr[k].reshape(-1)).max()/max(1e-12,np.abs(gr[k]).max())) for k,p in m.named_parameters()}
    worst=max(errs,key=errs.get)
    print(step,"logit err vs oracle-same-params %.2e"%np.abs(pred.detach().cpu().numpy()-lg).max(),"vs ref-traj %.2e"%np.abs(pred.detach().cpu().numpy()-g.z["steps/logits"][step-1]).max(),"worst rel grad:",worst,"%.2e"%errs[worst], " W-grad %.2e"%errs["linears.0.weight"])
    if step == 12:
        for k_,v_ in errs.items(): print('      ',k_,'%.2e'%v_)
        gw=m.linears[0].weight.grad.cpu().numpy(); np.set_printoptions(precision=4,linewidth=200); print((gw-gr['linears.0.weight'])[0]); print('alpha-ish gamma1 unit0', m.linears[1].weight[0].item())
    if step in (11,12,13):
        y2=cache['y2']; k2=np.unravel_index(np.argmin(np.abs(y2)),y2.shape); print('   min |y2| = %.3e at'%np.abs(y2).min(), k2, ' min|y3| %.3e'%np.abs(cache['y3']).min())
        e=cache['e']; n=cache['n']; B,U,_=e.shape
        ew=e[:,:,:7*n].reshape(B,U,n,7); srt=np.sort(ew,axis=3)
        gap=(srt[...,-1]-srt[...,-2])/srt[...,-1]
        k=np.unravel_index(np.argmin(gap),gap.shape)
        dq_dy=None
        print('   min rel gap between top-2 in a pooling window: %.3e at (b,u,w)=%s; windows with gap<1e-6: %d'%(gap.min(),k,(gap<1e-6).sum()))
        gw=m.linears[0].weight.grad.cpu().numpy(); d=np.abs(gw-gr['linears.0.weight']); ku=np.unravel_index(np.argmax(d),d.shape); print('   worst W-grad element',ku,'ours %.4e oracle %.4e'%(gw[ku],gr['linears.0.weight'][ku]), 'unit row max err per unit', d.reshape(U,-1).max(1))
    opt.step()

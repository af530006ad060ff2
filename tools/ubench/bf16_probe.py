import sys; sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tests']
import torch, yunet_oracle as O, yunet_amd, yunet_amd.synthetic as S
DEV='cuda'
def model(kind, sd, prec):
    cfg = yunet_amd.Config.fromfile(f'/root/repo/configs/yunet_{kind}.py')
    m = yunet_amd.build_detector(cfg.model); m.load_state_dict(sd, strict=True); m.to(DEV).train(); m.set_precision(prec); return m
for kind,h,n in (('n',320,16),('s',160,8),('n',320,64)):
    arch=O.yunet_arch(kind); sd=O.init_state(arch, seed=21); b=S.make_batch(n,h,h,77); out={}
    for prec in ('fp32','bf16'):
        m=model(kind,sd,prec); losses=m.forward_train(**S.to_device(b,DEV)); sum(losses.values()).backward(); torch.cuda.synchronize()
        p=m.engine.plan
        out[prec]=dict(l={k:float(v) for k,v in losses.items()}, gi=p.gt_inds.cpu().clone(), g=m.engine.params.grad.detach().cpu().clone(), flat=p.flat.cpu().clone(), ovl=p.max_overlaps.cpu().clone())
    a,c=out['fp32'],out['bf16']
    pos=(a['gi']>0)|(c['gi']>0); agree=float(((a['gi']==c['gi'])&pos).sum())/int(pos.sum())
    npa, npc = int((a['gi']>0).sum()), int((c['gi']>0).sum())
    cos=float((a['g']*c['g']).sum()/(a['g'].norm()*c['g'].norm()))
    print(kind,h,n,'agree',round(agree,4),'npos',npa,npc,'cos',round(cos,5),'flat rel',float((c['flat']-a['flat']).abs().max()/a['flat'].abs().max()))
    print('  fp32',a['l']); print('  bf16',c['l'])
    print('  mean matched iou fp32', float(a['ovl'][a['gi']>0].mean()), 'bf16', float(c['ovl'][c['gi']>0].mean()))

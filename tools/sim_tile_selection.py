"""Host replay of the tile-fed grad_value kernel's selection: for the bench's model-like encoder
locations (tools/kbench.hip, dist M) and the kernel's unit grid (gv_level_grid), how many samples does a unit decode for every
sample that lands in it, how many chunks does that make per (batch, head), and what would per-sample masks or larger units
change?  numpy only; `python tools/sim_tile_selection.py`."""
import numpy as np, sys
def grid(H,W,units_min=2,rows_max=256,MINW=64,BW=32):
    if W<MINW and W<=rows_max: nbx=1;bw=W
    else:
        target = BW if H>=rows_max//BW else rows_max//H
        nbx=-(-W//target); bw=-(-W//nbx); nbx=-(-W//bw)
    bh=max(1,min(H,rows_max//bw)); nby=-(-H//bh)
    if nbx*nby<units_min and nby<H:
        want=min(H,-(-units_min//nbx)); bh=-(-H//want); nby=-(-H//bh)
    return nbx,nby,bw,bh
def run(levels, m=1, seed=0, tileq=4, rows_max=256, BW=32, cellw=32, cellh=8):
    rng=np.random.default_rng(seed)
    S=sum(h*w for h,w in levels); Lq=S
    # reference points
    rx=np.concatenate([((np.arange(h*w)%w)+0.5)/w for h,w in levels]); ry=np.concatenate([((np.arange(h*w)//w)+0.5)/h for h,w in levels])
    th=m*2*np.pi/8; dx,dy=np.cos(th),np.sin(th); mx=max(abs(dx),abs(dy)); dx/=mx; dy/=mx
    tot=dict(box_tiles=0,mask_tiles=0,mask_queries=0,land=0,units=0, chunksA=0, chunksB=0, chunks0=0, exact_chunks=0)
    for l,(H,W) in enumerate(levels):
        k=np.arange(4)
        x=rx[:,None]+(dx*(k+1)[None,:]+rng.standard_normal((Lq,4)))/W
        y=ry[:,None]+(dy*(k+1)[None,:]+rng.standard_normal((Lq,4)))/H
        h=y*H-0.5; w=x*W-0.5
        inside=(h>-1)&(w>-1)&(h<H)&(w<W)
        h0=np.floor(h).astype(int); w0=np.floor(w).astype(int)
        xlo=np.where(w0>=0,w0,w0+1); xhi=np.where(w0+1<=W-1,w0+1,w0); ylo=np.where(h0>=0,h0,h0+1); yhi=np.where(h0+1<=H-1,h0+1,h0)
        nt=-(-Lq//tileq)
        pad=nt*tileq-Lq
        def padq(a,fill): return np.concatenate([a,np.full((pad,4),fill)]).reshape(nt,tileq*4)
        ins=padq(inside,False)
        big=10**9
        XL=padq(np.where(inside,xlo,big),big); XH=padq(np.where(inside,xhi,-big),-big); YL=padq(np.where(inside,ylo,big),big); YH=padq(np.where(inside,yhi,-big),-big)
        bxl=XL.min(1); bxh=XH.max(1); byl=YL.min(1); byh=YH.max(1)
        nbx,nby,bw,bh=grid(H,W,rows_max=rows_max,BW=BW)
        for by in range(nby):
            for bx in range(nbx):
                x0=bx*bw;x1=min(W,x0+bw);y0=by*bh;y1=min(H,y0+bh)
                boxhit=(bxl<x1)&(bxh>=x0)&(byl<y1)&(byh>=y0)
                # exact landing per sample: any corner in rect
                land=ins&(XL<x1)&(XH>=x0)&(YL<y1)&(YH>=y0)
                # cell masks: sample touches cell set; unit overlaps cells
                cx0,cx1=x0//cellw,(x1-1)//cellw; cy0,cy1=y0//cellh,(y1-1)//cellh
                cm=ins&((XL//cellw)<=cx1)&((XH//cellw)>=cx0)&((YL//cellh)<=cy1)&((YH//cellh)>=cy0)
                # big tiles (box spans >2 cells): all ones
                bigt=((bxh//cellw-bxl//cellw)>1)|((byh//cellh-byl//cellh)>1)
                cm=np.where(bigt[:,None],True,cm)&boxhit[:,None]
                mt=cm.any(1)
                mq=cm.reshape(nt,tileq,4).any(2)
                tot['box_tiles']+=boxhit.sum(); tot['mask_tiles']+=mt.sum(); tot['mask_queries']+=mq.sum(); tot['land']+=land.sum(); tot['units']+=1
                tot['chunks0']+=-(-boxhit.sum()*tileq//128); tot['chunksA']+=-(-mt.sum()*tileq//128); tot['chunksB']+=-(-mq.sum()//128)
                lq=land.reshape(nt,tileq,4).any(2).sum(); tot['exact_chunks']+=max(-(-lq//128), -(-land.sum()//512))
    return tot,Lq
for name,lv in (('360p',[(48,80),(24,40),(12,20),(6,10)]),('720p',[(92,160),(46,80),(23,40),(12,20)])):
    for m in (0,1):
        t,Lq=run(lv,m)
        print(name,'head',m,{k:int(v) for k,v in t.items()},'samples/level',Lq*4,'decoded0',t['box_tiles']*16,'x',round(t['box_tiles']*16/(4*Lq*4),2),'A',round(t['mask_tiles']*16/(16*Lq),2),'B',round(t['mask_queries']*4/(16*Lq),2), 'land',round(t['land']/(16*Lq),2))
print('--- bigger units')
for rm,bw in ((256,32),(512,32),(512,64),(1024,64)):
    for name,lv in (('360p',[(48,80),(24,40),(12,20),(6,10)]),('720p',[(92,160),(46,80),(23,40),(12,20)])):
        t,Lq=run(lv,1,rows_max=rm,BW=bw)
        print(rm,bw,name,'units',t['units'],'chunks0',t['chunks0'],'decoded x',round(t['box_tiles']*16/(16*Lq),2),'exact_chunks',t['exact_chunks'])

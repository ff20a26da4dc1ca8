"""A round-based model of one LM launch (round 4): 1024 SIMDs x 3 resident waves of 64 lanes, a work
queue with refills at >= 8 idle lanes, iteration counts sampled from the measured distribution of the 4-camera rig
(tools/lm_iter_hist.py), a SIMD's time per round = its alive waves x instructions per pass x 4 cycles.  Question: what would
merging the survivors of sparse waves (groups of four waves) buy?  Answer of the model: ~1 % - the launch is as long as
the SIMD that hosts a 1000-iteration system.  usage: python tools/experiments/lm_tail_model.py [n_sys] [instr per pass]"""
import numpy as np, sys
rng=np.random.default_rng(1)
def sample_iters(n, pts):
    # pts: list of (quantile, iters) ; inverse-CDF sampling with log-linear interpolation of iters vs -log(1-q)
    q=np.array([p[0] for p in pts]); v=np.array([p[1] for p in pts],float)
    u=rng.random(n)
    x=-np.log1p(-u); xq=-np.log1p(-q)
    it=np.interp(x, xq, v)
    return np.clip(np.round(it),3,1000).astype(int)
def simulate(n_sys, iters, instr_pass, waves_per_simd=3, n_simd=1024, merge=0, thr=8, coop_at=16):
    # event-free time stepping per "round": every alive wave executes one pass; a SIMD's round time = (#alive waves on it) * instr*4 cycles
    W=waves_per_simd*n_simd
    lanes=np.full((W,64),-1,int)       # remaining passes per lane (-1 idle)
    q=0
    order=iters
    t_simd=np.zeros(n_simd)
    simd_of=np.arange(W)%n_simd
    alive=np.ones(W,bool)
    total_cycles=0.0
    grp=np.arange(W)//4 if merge else None
    rounds=0
    while True:
        # refill
        idle=(lanes<0)
        nidle=idle.sum(1)
        if q<n_sys:
            need=np.where((nidle>=thr)&alive)[0]
            for w in need:
                k=min(nidle[w], n_sys-q)
                if k<=0: break
                idx=np.where(lanes[w]<0)[0][:k]
                lanes[w,idx]=order[q:q+k]; q+=k
        busy=(lanes>0).sum(1)
        alive=busy>0
        if not alive.any() and q>=n_sys: break
        if merge and q>=n_sys:
            # within each group of 4 waves: donors (busy<=16) give to a wave with more busy lanes and enough idle
            for g in np.unique(grp[alive]):
                ws=np.where((grp==g)&alive)[0]
                if len(ws)<2: continue
                b=busy[ws]
                o=ws[np.argsort(b)]
                for d in o:
                    if busy[d]==0 or busy[d]>16: continue
                    for tgt in o[::-1]:
                        if tgt==d or busy[tgt]==0: continue
                        if busy[tgt]>=busy[d] and 64-busy[tgt]>=busy[d]:
                            vals=lanes[d][lanes[d]>0]
                            idx=np.where(lanes[tgt]<=0)[0][:len(vals)]
                            lanes[tgt,idx]=vals; lanes[d]=-1
                            busy[tgt]+=len(vals); busy[d]=0
                            break
            alive=busy>0
        # one pass for every alive wave; time = max over SIMDs of (alive waves on SIMD)*instr*4
        cnt=np.bincount(simd_of[alive], minlength=n_simd)
        # all SIMDs proceed in parallel; approximate global round time by mean of busy SIMDs' time (waves are async in reality)
        t_simd+=cnt*instr_pass*4
        total_cycles=t_simd.max()
        lanes[lanes>0]-=1
        lanes[lanes==0]=-1
        rounds+=1
    return total_cycles/2.4e9*1e3, rounds
pts4=[(0,3),(0.5,34),(0.9,82),(0.99,271),(0.999,678),(0.9999,1000),(1.0,1000)]
n=int(sys.argv[1]) if len(sys.argv)>1 else 1000000
instr=int(sys.argv[2]) if len(sys.argv)>2 else 390
it=sample_iters(n, pts4)
print("mean iters",it.mean(),"cap",(it>=1000).sum())
for merge in (0,1):
    ms,r=simulate(n,it,instr,merge=merge)
    print("merge",merge,"ms",round(ms,3),"rounds",r)

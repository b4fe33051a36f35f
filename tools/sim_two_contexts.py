#!/usr/bin/env python3
"""Wave-level cost model of the decoupled render kernel with a second path context per lane.

Question: would a lane that keeps TWO paths (one in the traversal unit, one waiting for or coming
out of shading) pay for its extra registers and moves?  The model counts wave instructions
(the kernel is bound by VALU issue): a block of node steps, a batch of leaf tests, a shading
phase and a scheduling decision cost what the instrumented kernel shows (100 per node step, 70,
520, 25); rays are drawn with 1-19 node steps and leaves in between, 2.9 rays per path.  With one
context the model gives 39 instructions per ray with 38 / 44 / 26 lanes per node step / shading
phase / leaf batch (measured: 35.4 and 37 / 40 / 26).  With two contexts node steps run with
48-51 lanes, and the instructions per ray fall to 34.5 if an exchange of contexts costs 40
instructions per block and to 37.1 if it costs 70 (17 registers moved, and the new ray's
traversal set-up inside the node loop): 5-12 % for 17 more live registers.

Measured later: such a kernel was built (same images, 109 VGPRs) and ran 10-20 % SLOWER than the
one-context kernel at every setting of its thresholds, so the model is too kind: it prices an
exchange and a ray's start per block, but those blocks run with the few lanes that have just
finished, and it leaves out the extra scheduling decisions.  The kernel was removed again.
"""
import numpy as np
rng = np.random.default_rng(1)
def new_ray():
    u = rng.random()
    if u < 0.30:   n_in = rng.integers(1, 4)
    elif u < 0.75: n_in = rng.integers(3, 9)
    else:          n_in = rng.integers(8, 20)
    n_leaf = rng.poisson(0.25 * n_in + 0.6)
    ev = ['I'] * n_in
    for _ in range(n_leaf):
        ev.insert(rng.integers(1, len(ev) + 1), 'L')
    return ev
P_CONT = 1 - 1 / 2.9
C_NODE, C_LEAF, C_SHADE, C_REGEN, C_SCHED = 100, 70, 520, 40, 25
def run(n_target=60000, min_ready=40, leaf_min=8, blocked_min=2, C_SWAP=60, UNROLL=2, ring=True, two=True):
    # A: traversal context: None (free) / list (traversing) ; doneA: finished, waiting for B to be free
    # B: shading context: None / 'READY' / list (WAITING ray)
    A = [None]*64; pos=[0]*64; pend=[False]*64; doneA=[False]*64; B=[None]*64
    cost=0; done=0
    st=dict(nb=0,nl=0,sh=0,shl=0,lf=0,lfl=0)
    def start(l, ev): A[l]=ev; pos[l]=0; pend[l]=False; doneA[l]=False
    def exchange(l):
        # A finished: its hit goes to B (must be free of READY); A takes B's waiting ray, or a camera ray
        nxt = B[l] if B[l] is not None else (new_ray() if ring else None)
        B[l] = 'READY'
        if nxt is None: A[l]=None; doneA[l]=False
        else: start(l, nxt)
    while done < n_target:
        if not two:  # one context: the finished lanes are shaded where they are
            for l in range(64):
                if doneA[l]: B[l] = 'READY'; A[l] = None; doneA[l] = False
        nsh = sum(1 for b in B if b == 'READY')
        if nsh:
            for l in range(64):
                if B[l] == 'READY':
                    done += 1
                    B[l] = new_ray() if rng.random() < P_CONT else None
            cost += C_SHADE; st['sh']+=1; st['shl']+=nsh
        moved=False
        for l in range(64):
            if doneA[l] and B[l] != 'READY': exchange(l); moved=True
            if A[l] is None:
                if B[l] is not None and B[l] != 'READY': start(l, B[l]); B[l]=None; moved=True
                else: start(l, new_ray()); moved=True
        if moved: cost += C_SWAP + C_REGEN*0.3
        while True:
            cost += C_SCHED
            trav=[A[l] is not None and not doneA[l] for l in range(64)]
            nready = sum(1 for b in B if b == 'READY') + (0 if two else sum(doneA))
            nblocked = sum(doneA) if two else 0
            if not any(trav) or nready >= min_ready or nblocked >= 24: break
            act=0
            for k in range(UNROLL):
                for l in range(64):
                    if not trav[l]: continue
                    ev=A[l]
                    if pos[l] < len(ev) and ev[pos[l]]=='I': pos[l]+=1; act+=1
                    if pos[l] < len(ev) and ev[pos[l]]=='L' and not pend[l]: pend[l]=True; pos[l]+=1
            cost += C_NODE*UNROLL; st['nb']+=UNROLL; st['nl']+=act
            npend=sum(1 for l in range(64) if trav[l] and pend[l])
            if npend:
                blocked=sum(1 for l in range(64) if trav[l] and pend[l] and pos[l] < len(A[l]) and A[l][pos[l]]=='L')
                inner=sum(1 for l in range(64) if trav[l] and pos[l] < len(A[l]) and A[l][pos[l]]=='I')
                if npend >= leaf_min or blocked >= blocked_min or inner == 0:
                    cost += C_LEAF; st['lf']+=1; st['lfl']+=npend
                    for l in range(64):
                        if trav[l] and pend[l]:
                            pend[l]=False
                            if pos[l] < len(A[l]) and A[l][pos[l]]=='L': pend[l]=True; pos[l]+=1
            sw=False
            for l in range(64):
                if trav[l] and pos[l] >= len(A[l]) and not pend[l]:
                    if two and B[l] != 'READY': exchange(l); sw=True
                    else: doneA[l]=True
            if sw: cost += C_SWAP
    print(f"{'two contexts' if two else 'one context '} min_ready={min_ready} swap={C_SWAP} unroll={UNROLL}: cost/ray {cost/done:.1f} | node lanes {st['nl']/st['nb']:.1f} shade lanes {st['shl']/st['sh']:.1f} leaf lanes {st['lfl']/max(st['lf'],1):.1f} | per ray node steps {st['nb']/done:.3f} shades {st['sh']/done:.4f} leaves {st['lf']/done:.4f}")
run(min_ready=40, two=False)
for sw in (40, 70):
    for mr in (32, 40, 48, 56):
        run(min_ready=mr, C_SWAP=sw)
run(min_ready=48, C_SWAP=70, UNROLL=4)

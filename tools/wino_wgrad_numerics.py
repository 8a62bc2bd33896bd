#!/usr/bin/env python3
"""fp32 error of the Winograd weight gradient, transpose of F(4,3) ("F(3,4)": three taps from four output gradients and six inputs) against
the transpose of F(2,3) and an fp64 gradient - simulated on the CPU with the kernels' accumulation structure (fp32 accumulator, two
products per MFMA step, chains of `chain` position pairs per slab, slab sums in fp32) BEFORE csrc/conv_wino.hip::conv_wino_wgrad4_kernel
was written (round 5).  Result on 2048-pair chains, C = 64, N = 256, L = 63: F(2,3) 0.9-1.0e-6 of the tensor's scale, F(3,4) 1.6-2.2e-6
(ReLU'd / zero-mean inputs).  usage: python tools/wino_wgrad_numerics.py   (numpy only, ~1 minute)"""
import numpy as np
rng = np.random.default_rng(0)
f32 = np.float32
# F(3,4): outputs t=0..2, filter = 4 dy values, input 6 x values
BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
G = np.array([[1/4,0,0,0],[-1/6,-1/6,-1/6,-1/6],[-1/6,1/6,-1/6,1/6],[1/24,1/12,1/6,1/3],[1/24,-1/12,1/6,-1/3],[0,0,0,1]],dtype=np.float64)
AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,1]],dtype=np.float64)
# check algebra: out[t] = sum_i g[i] d[t+i]
g = rng.standard_normal(4); d = rng.standard_normal(6)
ref = np.array([sum(g[i]*d[t+i] for i in range(4)) for t in range(3)])
got = AT @ ((G@g)*(BT@d))
print("algebra", np.abs(ref-got).max())

def run(C=64, N=256, L=63, chain=2048, relu_x=True):
    dy = (rng.standard_normal((N, C, L))*0.01).astype(f32)
    x = rng.standard_normal((N, C, L)).astype(f32)
    if relu_x: x = np.maximum(x*1.0+0.3, 0).astype(f32)
    xp = np.pad(x, ((0,0),(0,0),(1,5)))
    # truth fp64
    truth = np.zeros((C, C, 3))
    for t in range(3):
        truth[:,:,t] = np.einsum('nol,nil->oi', dy.astype(np.float64), xp[:,:,t:t+L].astype(np.float64))
    scale = np.abs(truth).max()
    # ---- F(2,3) pairs
    Lh = (L+1)//2
    dyp = np.pad(dy, ((0,0),(0,0),(0,2*Lh-L)))
    e0 = dyp[:,:,0::2]; e1 = dyp[:,:,1::2]
    E = np.stack([e0, e0+e1, e0-e1, -e1]).astype(f32)    # 4,N,C,Lh
    d = [xp[:,:,i:i+2*Lh:2] for i in range(4)]
    V = np.stack([d[0]-d[2], d[1]+d[2], d[2]-d[1], d[1]-d[3]]).astype(f32)
    def chains(E, V, chain):
        P = E.shape[0]
        Ef = E.transpose(0,2,1,3).reshape(P, C, -1); Vf = V.transpose(0,2,1,3).reshape(P, C, -1)
        K = Ef.shape[2]
        tot = np.zeros((P, C, C), dtype=f32)
        nsl = 0
        for k0 in range(0, K, chain):
            acc = np.zeros((P, C, C), dtype=f32)
            for k in range(k0, min(K, k0+chain), 2):       # mfma 32x32x2: two products per accumulate
                pr = np.einsum('pok,pik->poi', Ef[:,:,k:k+2].astype(np.float64), Vf[:,:,k:k+2].astype(np.float64))
                acc = (acc.astype(np.float64) + pr).astype(f32)
            tot = (tot + acc).astype(f32); nsl += 1
        return tot
    M = chains(E, V, chain)
    hs = (M[1]+M[2])*f32(0.5); hd = (M[1]-M[2])*f32(0.5)
    dw2 = np.stack([M[0]+hs, hd, hs+M[3]], axis=-1)
    print("F(2,3) err/scale", np.abs(dw2-truth).max()/scale)
    # ---- F(3,4) quads
    Lq = (L+3)//4
    dyq = np.pad(dy, ((0,0),(0,0),(0,4*Lq-L)))
    e = [dyq[:,:,i::4] for i in range(4)]
    G32 = G.astype(f32)
    E6 = np.stack([sum(G32[r,i]*e[i] for i in range(4) if G[r,i]!=0) for r in range(6)]).astype(f32)
    dd = [xp[:,:,i:i+4*Lq:4] for i in range(6)]
    BT32 = BT.astype(f32)
    V6 = np.stack([sum(BT32[r,i]*dd[i] for i in range(6) if BT[r,i]!=0) for r in range(6)]).astype(f32)
    M6 = chains(E6, V6, chain//2)
    AT32 = AT.astype(f32)
    dw4 = np.stack([sum(AT32[t,r]*M6[r] for r in range(6) if AT[t,r]!=0) for t in range(3)], axis=-1).astype(f32)
    print("F(3,4) err/scale", np.abs(dw4-truth).max()/scale, "  rms", np.sqrt(((dw4-truth)**2).mean())/scale, " (F23 rms", np.sqrt(((dw2-truth)**2).mean())/scale, ")")
    # direct fp32 chain
run()
run(relu_x=False)

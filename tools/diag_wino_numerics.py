"""Offline (CPU, numpy) study for the next round: fp32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and a direct fp32 accumulation,
emulating the kernel's arithmetic (filters transformed in f64 and stored f32, input / output transforms and 4-channel MFMA-style accumulation
in f32) on post-ReLU activations with He-scaled weights; reference = direct convolution in f64."""
import numpy as np
rng=np.random.default_rng(0)
def mats(kind):
    if kind=='F2':
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],float)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],float)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],float)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],float)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],float)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],float)
    return BT,G,AT
def run(kind,C,K,ntiles=64,act='relu'):
    BT,G,AT=mats(kind); m=AT.shape[0]; a=BT.shape[0]
    # activations like post-ReLU features (non-negative, O(1)); weights He-scaled
    d=rng.standard_normal((ntiles,C,a,a)); d=np.maximum(d,0) if act=='relu' else d
    g=rng.standard_normal((K,C,3,3))*np.sqrt(2/(9*C))
    # reference f64 direct
    ref=np.zeros((ntiles,K,m,m))
    for i in range(m):
        for j in range(m):
            ref[:,:,i,j]=np.einsum('tcxy,kcxy->tk',d[:,:,i:i+3,j:j+3],g)
    f32=np.float32
    d32=d.astype(f32); 
    U=(G@g@G.T).astype(f32)                      # packed once (computed in f64, stored f32)
    V=np.einsum('ia,tcab->tcib',BT.astype(f32),d32).astype(f32); V=np.einsum('tcib,jb->tcij',V,BT.astype(f32)).astype(f32)
    # channel accumulation in fp32, sequential-ish (blocks of 4 like MFMA k=4)
    M=np.zeros((ntiles,K,a,a),f32)
    for c0 in range(0,C,4):
        M+=np.einsum('tcij,kcij->tkij',V[:,c0:c0+4],U[:,c0:c0+4]).astype(f32)
    Y=np.einsum('ia,tkab->tkib',AT.astype(f32),M).astype(f32); Y=np.einsum('tkib,jb->tkij',Y,AT.astype(f32)).astype(f32)
    err=np.abs(Y-ref).max(); return err, np.abs(ref).max(), err/np.abs(ref).max()
for C,K in ((64,64),(128,128),(256,256)):
    for kind in ('F2','F4'):
        e=run(kind,C,K)
        print(kind,C,K,'max abs err %.3e  max|ref| %.2f  rel-to-max %.2e'%e)
# direct fp32 accumulation for comparison
def direct32(C,K,ntiles=64):
    d=np.maximum(rng.standard_normal((ntiles,C,3,3)),0); g=rng.standard_normal((K,C,3,3))*np.sqrt(2/(9*C))
    ref=np.einsum('tcxy,kcxy->tk',d,g)
    acc=np.zeros((ntiles,K),np.float32)
    for c0 in range(0,C,4):
        for x in range(3):
            for y in range(3):
                acc+=np.einsum('tc,kc->tk',d[:,c0:c0+4,x,y].astype(np.float32),g[:,c0:c0+4,x,y].astype(np.float32)).astype(np.float32)
    return np.abs(acc-ref).max()/np.abs(ref).max()
print('direct fp32 rel-to-max', direct32(64,64), direct32(256,256))

import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd import ops
def T(fn, it=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
for (M,K,N) in [(962,300,300),(962,300,600)]:
    A=torch.randn(M,K,device="cuda"); B=torch.randn(K,N,device="cuda"); G=torch.randn(M,N,device="cuda")
    C=torch.empty(M,N,device="cuda")
    from jmac_amd._lib import lib, ptr, stream
    L=lib(); st=stream()
    print("M=%d K=%d N=%d" % (M,K,N))
    print("  fwd NN: jmac %.1f us (abi %.1f) torch %.1f us" % (T(lambda: ops._gemm(A,False,B,False,M,N,K)), T(lambda: L.jmac_gemm_f32(ptr(A),K,0,ptr(B),N,0,M,N,K,ptr(C),N,st)), T(lambda: A@B)))
    dA=torch.empty(M,K,device="cuda"); dB=torch.empty(K,N,device="cuda")
    print("  dA NT : jmac abi %.1f us torch %.1f us" % (T(lambda: L.jmac_gemm_f32(ptr(G),N,0,ptr(B),N,1,M,K,N,ptr(dA),K,st)), T(lambda: G@B.t())))
    print("  dB TN : jmac abi %.1f us torch %.1f us" % (T(lambda: L.jmac_gemm_f32(ptr(A),K,1,ptr(G),N,0,K,N,M,ptr(dB),N,st)), T(lambda: A.t()@G)))

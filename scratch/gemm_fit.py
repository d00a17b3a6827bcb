import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmac_amd import scoring
from jmac_amd._lib import lib, ptr, stream
L=lib(); st=stream()
for n in (1024, 2048, 4096, 6016, 8192, 12032, 16384, 24064):
    A=torch.randn(n,300,device="cuda"); B=torch.randn(n,300,device="cuda"); C=torch.empty(n,n,device="cuda")
    fn=lambda: L.jmac_sim_matrix_f32(ptr(A),300,ptr(B),300,n,n,300,ptr(C),n,st)
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); t=e0.elapsed_time(e1)/10*1e3
    for _ in range(3): A@B.t()
    e0.record()
    for _ in range(10): torch.mm(A,B.t(),out=C)
    e1.record(); torch.cuda.synchronize(); t2=e0.elapsed_time(e1)/10*1e3
    print("n=%6d tiles=%6d jmac %8.1f us %6.1f TF | torch %8.1f us %6.1f TF" % (n,((n+127)//128)**2,t,2*n*n*300/t/1e6,t2,2*n*n*300/t2/1e6))
n=12000
A=torch.randn(n,300,device="cuda"); B=torch.randn(n,300,device="cuda"); C=torch.empty(n,n,device="cuda")
def T(fn):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/10*1e3
print("n=12000 out= : %.1f us; alloc each call: %.1f us; torch alloc: %.1f us; torch out=: %.1f us" % (T(lambda: scoring.sim_matrix(A,B,out=C)), T(lambda: scoring.sim_matrix(A,B)), T(lambda: A@B.t()), T(lambda: torch.mm(A,B.t(),out=C))))
tab=torch.nn.functional.normalize(torch.randn(30000,300,device="cuda"))
A2,B2=tab[:12000],tab[12000:24000]
print("normalized slices: %.1f us (first) %.1f us (again)" % (T(lambda: scoring.sim_matrix(A2,B2,out=C)), T(lambda: scoring.sim_matrix(A2,B2,out=C))))
A3=torch.randn(12000,300,device="cuda")*0.058; B3=torch.randn(12000,300,device="cuda")*0.058
print("small randn separate: %.1f us" % T(lambda: scoring.sim_matrix(A3,B3,out=C)))
print("big randn again: %.1f us" % T(lambda: scoring.sim_matrix(A,B,out=C)))

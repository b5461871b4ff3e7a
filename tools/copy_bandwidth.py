import sys, torch
N=4096
a=torch.randn(N+6,N+6,dtype=torch.float64,device='cuda'); b=torch.randn_like(a); c=torch.empty_like(a); d=torch.empty_like(a)
def timeit(fn,n=100):
    for _ in range(300): fn()   # (device clocks settle after ~30 ms of load)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
t=timeit(lambda:(c.copy_(a),d.copy_(b)))
print(f"2x torch copy: {t*1e3:.1f} us -> {4*a.numel()*8/t/1e6:.0f} GB/s")
t=timeit(lambda:torch.add(a,b,out=c))
print(f"torch add (2R 1W): {t*1e3:.1f} us -> {3*a.numel()*8/t/1e6:.0f} GB/s")
big=torch.randn(1<<28,dtype=torch.float32,device='cuda'); big2=torch.empty_like(big)
t=timeit(lambda:big2.copy_(big))
print(f"1 GiB f32 copy: {t*1e3:.1f} us -> {2*big.numel()*4/t/1e6:.0f} GB/s")
ai=a[3:-3,3:-3]; ci=c[3:-3,3:-3]
t=timeit(lambda:ci.copy_(ai))
print(f"strided interior copy: {t*1e3:.1f} us -> {2*ai.numel()*8/t/1e6:.0f} GB/s")

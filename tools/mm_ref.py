"""What the vendor library (hipBLASLt through torch.mm) reaches on the stage 2-3 product shapes, plain GEMM without the
statistics epilogue: a yardstick for conv1x1_gemm_kernel, not a path of this package."""
import torch
shapes = [(256,256,524288),(1024,256,524288),(256,1024,524288),(256,512,524288),(512,256,524288),(256,1280,524288),
          (512,512,262144),(2048,512,262144),(512,2048,262144),(512,1024,262144),(1024,512,262144),(512,2560,262144),
          (512,128,1048576),(128,512,1048576)]
dev = torch.device("cuda")
for R,K,M in shapes:
    W = torch.randn(R,K,device=dev,dtype=torch.bfloat16); X = torch.randn(K,M,device=dev,dtype=torch.bfloat16)
    Y = torch.empty(R,M,device=dev,dtype=torch.bfloat16)
    for _ in range(3): torch.mm(W,X,out=Y)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): torch.mm(W,X,out=Y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)*100
    fl = 2.0*R*K*M; by=(R+K)*M*2
    print(f"R={R:5d} K={K:5d} M={M:8d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e6:6.2f} TB/s  floor {max(fl/2.5e9, by/8e6):7.1f} us", flush=True)

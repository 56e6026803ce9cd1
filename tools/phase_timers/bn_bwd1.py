import subprocess, glob, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "grafp_amd/csrc/bn.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b, 1)
i = s.index('__global__ __launch_bounds__(BN1_THREADS) void bn_bwd1_kernel(')
head, tail = s[:i], s[i:]
def rept(a, b):
    global tail
    assert tail.count(a) == 1, a
    tail = tail.replace(a, b, 1)
rept('''    typename BnIO<T>::Raw rx[ITEMS], rd[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
        if (m < hi) {
            rx[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(row + m);
            rd[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(grow + m);
        }
    }
    float sd = 0.0f, sdx = 0.0f;''', '''    typename BnIO<T>::Raw rx[ITEMS], rd[ITEMS];
    unsigned long long TT[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk1;
#define TICK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tk1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TT[i] += tk1 - tk0; tk0 = tk1; } while (0)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t m = lo + ((int64_t)it * BN1_THREADS + tid) * W;
        if (m < hi) {
            rx[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(row + m);
            rd[it] = *reinterpret_cast<const typename BnIO<T>::Raw *>(grow + m);
        }
    }
    TICK(0);
    float sd = 0.0f, sdx = 0.0f;''')
rept('''    const float2 r = block_sum2<BN1_THREADS>(sd, sdx, scratch, tid);
    unsigned long long *slots_row''', '''    const float2 r = block_sum2<BN1_THREADS>(sd, sdx, scratch, tid);
    TICK(1);
    unsigned long long *slots_row''')
rept('''#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        bn_opaque(rx[it]);
        bn_opaque(rd[it]);
    }
    int checkout = 0;''', '''    TICK(2);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        bn_opaque(rx[it]);
        bn_opaque(rd[it]);
    }
    int checkout = 0;''')
rept('''            BnIO<T>::store(orow + m, v);
        }
    }
    if (tid == 0) bn1_rearm(checkout, S, counter, slots_row);
}

// single-pass plan''', '''            BnIO<T>::store(orow + m, v);
        }
    }
    TICK(3);
    if (tid == 0) bn1_rearm(checkout, S, counter, slots_row);
    if (blockIdx.y == 7 && (blockIdx.x == 3 || blockIdx.x == S - 2) && tid == 0 && S >= 16)
        printf("bn_bwd1 ITEMS %d S %d s %d: loads %llu reduce %llu rendezvous %llu compute+stores(drained) %llu\\n", ITEMS, S, s, TT[0], TT[1], TT[2], TT[3]);
}

// single-pass plan''')
open("/tmp/bn_tm.hip", "w").write(head + tail)
csrc = os.path.join(root, "grafp_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(root, "include"), "-I" + csrc, "-Wno-inline-asm", "-c", "/tmp/bn_tm.hip", "-o", "/tmp/bn_tm.o"])
objs = [o for o in glob.glob(os.path.join(csrc, "_obj/*.o")) if not o.endswith("/bn.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["/tmp/bn_tm.o", "-o", os.path.join(root, "scratch/lib_tm_bn.so")])

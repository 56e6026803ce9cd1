# builds scratch/lib_tm_wg.so: wgrad.hip with s_memtime segment timers in wgrad_dma_kernel (printf from one workgroup)
import subprocess, glob, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "grafp_amd/csrc/wgrad.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b, 1)
rep('''    for (int t = 0; t < T; ++t) {
        // chunks t .. min(t + D, T) - 1 are in flight: allow all but chunk t's
        const int ahead = (T - 1 - t < D - 1) ? T - 1 - t : D - 1;
        gm_wait_allowed(__builtin_amdgcn_readfirstlane(ahead * DMA_PER_CHUNK));
        __builtin_amdgcn_s_barrier();
        if (t + D < T) issue(t + D);
        unsigned char *const st = smem + (t % NS) * CFG::STAGE;''', '''    unsigned long long TT[4] = {0, 0, 0, 0};
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk1;
#define TICK(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TT[i] += tk1 - tk0; tk0 = tk1; } while (0)
    for (int t = 0; t < T; ++t) {
        // chunks t .. min(t + D, T) - 1 are in flight: allow all but chunk t's
        const int ahead = (T - 1 - t < D - 1) ? T - 1 - t : D - 1;
        gm_wait_allowed(__builtin_amdgcn_readfirstlane(ahead * DMA_PER_CHUNK));
        TICK(0);
        __builtin_amdgcn_s_barrier();
        TICK(1);
        if (t + D < T) issue(t + D);
        TICK(2);
        unsigned char *const st = smem + (t % NS) * CFG::STAGE;''')
rep('''                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // partial tile -> part[slice][grp][o][c]''', '''                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        { float sink = acc[0][0][0]; asm volatile("" :: "v"(sink)); }
        TICK(3);
    }
    if (blockIdx.x == 300 && blockIdx.z == 0 && lane == 0 && T > 8)
        printf("w%d T %d (tile %dx%d NS %d): vmwait %llu barrier %llu issue %llu compute %llu per chunk\\n", wave, T, CFG::TO, CFG::TC, NS, TT[0] / T, TT[1] / T, TT[2] / T, TT[3] / T);
    // partial tile -> part[slice][grp][o][c]''')
open("/tmp/wgrad_tm.hip", "w").write(s)
csrc = os.path.join(root, "grafp_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(root, "include"), "-I" + csrc, "-Wno-inline-asm", "-c", "/tmp/wgrad_tm.hip", "-o", "/tmp/wg_tm.o"])
objs = [o for o in glob.glob(os.path.join(csrc, "_obj/*.o")) if not o.endswith("/wgrad.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["/tmp/wg_tm.o", "-o", os.path.join(root, "scratch/lib_tm_wg.so")])

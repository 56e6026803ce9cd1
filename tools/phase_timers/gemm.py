import subprocess, glob, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "grafp_amd/csrc/gemm.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b, 1)
rep('''    int ch = 0, tile = 0;
    for (int t = 0; t < T; ++t) {''', '''    int ch = 0, tile = 0;
    unsigned long long TT[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk1;
#define TICK(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tk1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TT[i] += tk1 - tk0; tk0 = tk1; } while (0)
    for (int t = 0; t < T; ++t) {''')
rep('''            gm_wait_allowed(__builtin_amdgcn_readfirstlane(allowed));
            __builtin_amdgcn_s_barrier();
        }''', '''            gm_wait_allowed(__builtin_amdgcn_readfirstlane(allowed));
            TICK(0);
            __builtin_amdgcn_s_barrier();
            TICK(1);
        }''')
rep('''        unsigned char *const st = smem + (t % NS) * CFG::STAGE;
        // ---- 2 k-steps x (2 x RT) MFMAs ----''', '''        TICK(2);
        unsigned char *const st = smem + (t % NS) * CFG::STAGE;
        // ---- 2 k-steps x (2 x RT) MFMAs ----''')
rep('''        int stored_now = 0;
        if (++ch == nch) {''', '''        { float sink = acc[0][0][0] + acc[1][RT - 1][15]; asm volatile("" :: "v"(sink)); }
        TICK(3);
        int stored_now = 0;
        if (++ch == nch) {''')
rep('''        // shift the issue history by one iteration
#pragma unroll
        for (int j = D - 1; j > 0; --j) dma_hist[j] = dma_hist[j - 1];''', '''        TICK(4);
        // shift the issue history by one iteration
#pragma unroll
        for (int j = D - 1; j > 0; --j) dma_hist[j] = dma_hist[j - 1];''')
rep('''    if (STATS) {
        // per row: this wave's (sum, sum of squares, shift) over its 64-column share of every tile of the range ->''', '''    if (blockIdx.x == 37 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0 && T >= 8)
        printf("gemm TR %d TN %d NW %d w%d T %d nch %d tiles %d STATS %d: per chunk: vmwait %llu barrier %llu issue %llu mfma %llu | epilogue per tile %llu\\n", CFG::TR, TN, CFG::NW, (int)(threadIdx.x >> 6), T, nch, tile, (int)STATS, TT[0] / T, TT[1] / T, TT[2] / T, TT[3] / T, TT[4] / (tile > 0 ? tile : 1));
    if (STATS) {
        // per row: this wave's (sum, sum of squares, shift) over its 64-column share of every tile of the range ->''')
open("/tmp/gemm_tm.hip", "w").write(s)
csrc = os.path.join(root, "grafp_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(root, "include"), "-I" + csrc, "-Wno-inline-asm", "-c", "/tmp/gemm_tm.hip", "-o", "/tmp/gemm_tm.o"])
objs = [o for o in glob.glob(os.path.join(csrc, "_obj/*.o")) if not o.endswith("/gemm.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["/tmp/gemm_tm.o", "-o", os.path.join(root, "scratch/lib_tm_gemm.so")])

# builds scratch/lib_tm.so: knn_search.hip with s_memtime segment timers in the bf16 tile loop (printf from one workgroup)
import subprocess, glob, os
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(root, "grafp_amd/csrc/knn_search.hip")).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b, 1)
rep('''    int stage = 0;
    for (int t = 0; t < ntiles; ++t) {
        gm_wait_vm<(SB_NS - 2) * SB_PER>();                   // this wave's part of tile t has landed
        __syncthreads();  // everybody's part has; every wave is done with tile t-1, whose stage is free again
        on_tile(t);       // quiescent point: no wave is inside on_block, so workgroup state is uniform here''',
'''    int stage = 0;
    unsigned long long T[8] = {0,0,0,0,0,0,0,0};
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), c1;
#define TICK(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); T[i] += c1 - c0; c0 = c1; } while (0)
    for (int t = 0; t < ntiles; ++t) {
        gm_wait_vm<(SB_NS - 2) * SB_PER>();                   // this wave's part of tile t has landed
        TICK(0);
        __syncthreads();  // everybody's part has; every wave is done with tile t-1, whose stage is free again
        TICK(1);
        on_tile(t);       // quiescent point: no wave is inside on_block, so workgroup state is uniform here
        TICK(2);''')
rep('''        issue(t + SB_NS - 1 < ntiles ? t + SB_NS - 1 : ntiles - 1, free_stage);
''', '''        issue(t + SB_NS - 1 < ntiles ? t + SB_NS - 1 : ntiles - 1, free_stage);
        TICK(3);
''')
rep('''            __builtin_amdgcn_sched_barrier(0);
            // NQS independent accumulator chains''', '''            __builtin_amdgcn_sched_barrier(0);
            TICK(4);
            // NQS independent accumulator chains''')
rep('''            __builtin_amdgcn_sched_barrier(0);
            float hv[16]; ''', '''            __builtin_amdgcn_sched_barrier(0);
            { float sink = 0; for (int j = 0; j < NQS; ++j) sink += acc[j][0]; asm volatile("" :: "v"(sink)); }
            TICK(5);
            float hv[16]; ''')
rep('''            for (int j = 0; j < NQS; ++j) on_block(t, rb, j, acc[j], hv);
        }''', '''            for (int j = 0; j < NQS; ++j) { const unsigned long long before = T[7]; on_block(t, rb, j, acc[j], hv, T[7]); if (T[7] != before) TICK(7); else TICK(6); }
        }''')
rep('''                               [&](int t, int rb, int j, const f32x16 &acc, const float (&hv)[16]) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = __builtin_fmaf(hv[r], -hs, acc[r]);''', '''                               [&](int t, int rb, int j, const f32x16 &acc, const float (&hv)[16], unsigned long long &slow) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e[r] = __builtin_fmaf(hv[r], -hs, acc[r]);''')
rep('''        if (__ballot(m >= a_q[j]) != 0) {
''', '''        if (__ballot(m >= a_q[j]) != 0) {
            slow += 1000000000ull;
''')
rep('''                               [&](int, int, int j, const f32x16 &acc, const float (&hv)[16]) {''', '''                               [&](int, int, int j, const f32x16 &acc, const float (&hv)[16], unsigned long long &) {''')
rep('''    gm_wait_vm<0>();      // nothing of this wave's is in flight when the caller reuses LDS or the wave ends''', '''    gm_wait_vm<0>();      // nothing of this wave's is in flight when the caller reuses LDS or the wave ends
    if (blockIdx.x == 5 && blockIdx.y == 3 && (tid & 63) == 0 && ntiles > 100)
        printf("w%d tiles %d: vmwait %llu barrier %llu on_tile %llu issue %llu ldsread %llu mfma %llu | fast-epi total %llu slow-epi count %llu total %llu\\n", wave, ntiles, T[0]/ntiles, T[1]/ntiles, T[2]/ntiles, T[3]/ntiles, T[4]/ntiles, T[5]/ntiles, T[6], T[7] / 1000000000ull, T[7] % 1000000000ull);''')
open("/tmp/knn_search_tm.hip", "w").write(s)
csrc = os.path.join(root, "grafp_amd/csrc")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-I" + os.path.join(root, "include"), "-I" + csrc, "-Wno-inline-asm", "-c", "/tmp/knn_search_tm.hip", "-o", "/tmp/ks_tm.o"])
objs = [o for o in glob.glob(os.path.join(csrc, "_obj/*.o")) if "knn_search" not in o]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["/tmp/ks_tm.o", "-o", os.path.join(root, "scratch/lib_tm.so")])

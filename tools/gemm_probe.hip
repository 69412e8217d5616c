// Timeline of the split-f16 GEMM workgroups (round 6, VERDICT r05 item 1: "or a timeline showing which condition fails").
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DGEMM_TIMELINE -I some_amd/csrc tools/gemm_probe.hip -o tools/_bin/gemm_probe
// Every workgroup's wavefronts record s_memtime at: kernel entry (t0), behind the prologue barrier - first k-block in LDS, second in
// registers - (t1), behind the last product (t2), behind the last epilogue store instruction (t3), behind s_waitcnt vmcnt(0) (t4), plus
// the CU they ran on.  Reported per shape / tile: where a CU's time goes (dispatch gap between consecutive workgroups of a CU,
// prologue, k-loop, epilogue issue, store drain), in microseconds (the counter is calibrated against the HIP event time of the launch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../some_amd/csrc/gemm_f16x3.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

static void fill_split(std::vector<uint16_t>& v, float scale, unsigned seed) {
    srand(seed);
    for (size_t i = 0; i < v.size(); i += 64) {
        for (int k = 0; k < 32; ++k) {
            const float x = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * scale;
            half_t h, l;
            split_f16(x, h, l);
            v[i + k] = __builtin_bit_cast(uint16_t, h);
            v[i + 32 + k] = __builtin_bit_cast(uint16_t, l);
        }
    }
}

int main(int argc, char** argv) {
    // row counts that make whole rounds of workgroups for both tiles (no wave-quantisation tail launch to tell apart)
    struct Shape { const char* name; GemmEpi epi; int N, K; bool out_split; int M; };
    const Shape shapes[] = {{"ffn1 512->2048 bias+SiLU, SPLIT32 out", EPI_BIAS_SILU, 2048, 512, true, 81920},
                            {"ffn2 2048->512 bias+residual", EPI_BIAS_RES, 512, 2048, false, 65536},
                            {"glu 512->2x512", EPI_GLU, 1024, 512, false, 81920}};
    (void)argc; (void)argv;
    const int cap = 8192;
    unsigned long long* tl_d;
    CK(hipMalloc(&tl_d, (size_t)cap * 8 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_tl), &tl_d, sizeof(tl_d)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_tl_cap), &cap, sizeof(cap)));
    unsigned long long* it_d;
    CK(hipMalloc(&it_d, 8 * 1024 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_it), &it_d, sizeof(it_d)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const int M = sh.M;
        std::vector<uint16_t> ha((size_t)M * sh.K * 2), hw((size_t)sh.N * sh.K * 2);
        fill_split(ha, 1.0f, 1);
        fill_split(hw, 0.05f, 2);
        float *A, *W, *bias, *C, *res;
        const int n_out = sh.epi == EPI_GLU ? sh.N / 2 : sh.N;
        CK(hipMalloc(&A, ha.size() * 2)); CK(hipMalloc(&W, hw.size() * 2)); CK(hipMalloc(&bias, sh.N * 4));
        CK(hipMalloc(&C, (size_t)M * n_out * 4)); CK(hipMalloc(&res, (size_t)M * n_out * 4));
        CK(hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, sh.N * 4)); CK(hipMemset(res, 0, (size_t)M * n_out * 4));
        for (int variant : {2, 5, 6}) {                  // 2: 256 x 256, one tile per workgroup; 5: single stage, two per CU; 6: persistent stream
            const int tile = variant == 6 ? 2 : variant;
            GemmArgs a{};
            a.g[0] = GemmGroup{A, W, bias, res, C, nullptr, sh.N, 0};
            a.groups = 1; a.M = M; a.K = sh.K; a.lda = sh.K; a.ldc = n_out; a.ldr = n_out; a.alpha = 0.5f;
            a.flags = GEMM_FLAG_TR | (variant == 6 ? GEMM_FLAG_PERSIST : 0);
            // STEADY STATE first: the package power limit sets the clock over ~milliseconds, so a lone launch behind an idle gap runs at
            // a boost clock a stream of launches never sees.  ~0.4 s of warm-up, ~1 s timed: `steady` is what a model step experiences.
            for (int i = 0; i < 3; ++i) CK(launch_gemm_f16x3(sh.epi, a, sh.out_split, tile, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            CK(launch_gemm_f16x3(sh.epi, a, sh.out_split, tile, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float one_ms = 0;
            CK(hipEventElapsedTime(&one_ms, e0, e1));
            const int reps = (int)(1000.f / one_ms) + 1;
            for (int i = 0; i < reps * 2 / 5; ++i) CK(launch_gemm_f16x3(sh.epi, a, sh.out_split, tile, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) CK(launch_gemm_f16x3(sh.epi, a, sh.out_split, tile, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float steady_ms = 0;
            CK(hipEventElapsedTime(&steady_ms, e0, e1));
            steady_ms /= reps;
            // the timeline below is taken from ONE more launch right behind that stream (no idle gap in front of it)
            CK(hipMemset(tl_d, 0, (size_t)cap * 8 * 8 * 8));
            CK(hipMemset(it_d, 0, 8 * 1024 * 8));
            { int zero[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_it_n), zero, sizeof(zero))); }
            CK(hipEventRecord(e0, 0));
            CK(launch_gemm_f16x3(sh.epi, a, sh.out_split, tile, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const int waves = tile == 2 ? 8 : 4;
            std::vector<unsigned long long> tl((size_t)cap * 8 * 8);
            CK(hipMemcpy(tl.data(), tl_d, tl.size() * 8, hipMemcpyDeviceToHost));
            // records of wave 0 of every workgroup of the MAIN launch (the tail launch of the wave-quantisation split overwrites the
            // first slots with its own, smaller, workgroups: told apart by a k-loop that is far shorter)
            struct Rec { unsigned long long t0, t1, t2, t3, t4; unsigned long long cu; unsigned long long skew; };
            std::vector<Rec> recs;
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int wg = 0; wg < cap; ++wg) {
                const unsigned long long* o = &tl[((size_t)wg * waves) * 8];
                if (o[0] == 0) continue;
                unsigned long long lo2 = ~0ull, hi2 = 0;
                for (int w = 0; w < waves; ++w) { const unsigned long long t2 = tl[((size_t)wg * waves + w) * 8 + 2]; lo2 = std::min(lo2, t2); hi2 = std::max(hi2, t2); }
                // HW_ID (gfx9): [11:8] cu_id, [12] sh_id, [15:13] se_id; XCC_ID [3:0]
                const unsigned long long key = ((o[6] & 0xf) << 16) | ((o[5] >> 8) & 0xff);
                recs.push_back(Rec{o[0], o[1], o[2], o[3], o[4], key, hi2 - lo2});
                tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[4]);
            }
            if (recs.empty()) { printf("%s variant %d: no records\n", sh.name, variant); continue; }
            // the counter's base differs from CU to CU (per-XCD clocks): calibrate on what one CU saw - from its first workgroup's entry to its
            // last workgroup's last store is (almost exactly) the launch as the HIP events timed it
            std::map<unsigned long long, std::pair<unsigned long long, unsigned long long>> cspan;
            for (const Rec& r : recs) {
                auto it = cspan.find(r.cu);
                if (it == cspan.end()) cspan[r.cu] = {r.t0, r.t4};
                else { it->second.first = std::min(it->second.first, r.t0); it->second.second = std::max(it->second.second, r.t4); }
            }
            double span = 0;
            for (auto& kv : cspan) span += (double)(kv.second.second - kv.second.first);
            const double ticks_launch = span / cspan.size();
            const double tick_us = ms * 1e3 / ticks_launch;
            (void)tmin; (void)tmax;
            std::map<unsigned long long, std::vector<Rec>> per_cu;
            for (const Rec& r : recs) per_cu[r.cu].push_back(r);
            double pro = 0, loop = 0, epi = 0, drain = 0, gap = 0, first = 0, skew = 0;
            size_t n = 0, ngap = 0;
            for (auto& kv : per_cu) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end(), [](const Rec& x, const Rec& y) { return x.t0 < y.t0; });
                first += 0.0;
                for (size_t i = 0; i < v.size(); ++i) {
                    pro += (double)(v[i].t1 - v[i].t0); loop += (double)(v[i].t2 - v[i].t1); epi += (double)(v[i].t3 - v[i].t2);
                    drain += (double)(v[i].t4 - v[i].t3); skew += (double)v[i].skew; ++n;
                    // next workgroup on the same CU slot: the first that starts after this one ends (two slots per CU for tile 5)
                    for (size_t j = i + 1; j < v.size(); ++j)
                        if (v[j].t0 >= v[i].t4) { gap += (double)(v[j].t0 - v[i].t4); ++ngap; break; }
                }
            }
            {   // k-block barrier stamps of the workgroup(s) with blockIdx.x == 8, wavefront 0: iteration lengths in counter ticks
                std::vector<unsigned long long> it(8 * 1024);
                CK(hipMemcpy(it.data(), it_d, it.size() * 8, hipMemcpyDeviceToHost));
                int cnt[8];
                CK(hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_gemm_it_n), sizeof(cnt)));
                printf("    k-block lengths (ticks) seen by wavefront 0 of blockIdx.x == 8, %d stamps:", cnt[0]);
                for (int i = 1; i < cnt[0] && i < 100; ++i) printf(" %llu", it[i] - it[i - 1]);
                printf("\n");
            }
            const double wg_us = (pro + loop + epi + drain) / n * tick_us;
            printf("%-40s %s: STEADY %.4f ms per launch (%d back to back); the traced launch %.4f ms = %.0f counter ticks on a CU (%.0f ticks per us); "
                   "%zu tiles on %zu CU ids, %.1f per CU id\n", sh.name, variant == 2 ? "256x256" : variant == 5 ? "128x256 single stage x 2" : "256x256 persistent",
                   steady_ms, reps, ms, ticks_launch, ticks_launch / (ms * 1e3), n, per_cu.size(), (double)n / per_cu.size());
            printf("    per workgroup (us): prologue %.2f | k-loop %.2f | epilogue to the last store issued %.2f | store drain %.2f | total %.2f | gap to "
                   "the next workgroup of the CU slot %.2f | last-product skew between the wavefronts %.2f %s\n",
                   pro / n * tick_us, loop / n * tick_us, epi / n * tick_us, drain / n * tick_us, wg_us, ngap ? gap / ngap * tick_us : 0.0,
                   skew / n * tick_us, first > 0 ? "" : "");
            printf("    share of a workgroup's residency: prologue %.1f %% | k-loop %.1f %% | epilogue %.1f %% | drain %.1f %%;  dispatch gap = %.1f %% on top\n",
                   100 * pro / (pro + loop + epi + drain), 100 * loop / (pro + loop + epi + drain), 100 * epi / (pro + loop + epi + drain),
                   100 * drain / (pro + loop + epi + drain), ngap ? 100 * (gap / ngap) / ((pro + loop + epi + drain) / n) : 0.0);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(C)); CK(hipFree(res));
    }
    return 0;
}

// Host-side check of the own-group layout's run tables (persist_device.hip.h: ps_own_runs / ps_own_remote / ps_wave_range), for the
// shapes the engine runs: every tile of [FFN2 | out-proj] is streamed exactly once, every column group is finished by exactly one
// workgroup, the waves' shares tile a workgroup's run space, and a merger's list of remote partials is exactly the (run, wave) pieces
// the other workgroups publish for its group.  tests/test_kernel_resources.py builds and runs it (no GPU needed).
#include "persist_device.hip.h"

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
using namespace ftcf;

static int check(int NB, int NG, int KT_a, int KT_b, int TK, int M, int Il, int cs3)
{
    if (!ps_own_ok(NB, NG, M)) {
        printf("NB %d NG %d: not eligible\n", NB, NG);
        return 0;
    }
    std::vector<std::vector<int>> covF(NG, std::vector<int>(KT_b, 0)), covO(NG, std::vector<int>(KT_a, 0));
    std::map<int, int>            fin;
    int                           maxT = 0, minT = 1 << 30, maxrem = 0;
    for (int b = 0; b < NB; b++) {
        RunRec    r[PS_RMAX], tmp[PS_RMAX];
        const int n = ps_own_runs(b, NB, NG, KT_a, KT_b, TK, M, Il, r);
        int       T = 0, merged = -1;
        for (int j = 0; j < n; j++) {
            T += r[j].nt;
            const int g = r[j].grp, t0 = r[j].tile0 - g * (r[j].sel ? KT_a : KT_b);
            for (int t = 0; t < r[j].nt; t++) {
                (r[j].sel ? covO : covF)[g][t0 + t]++;
            }
            if (!(r[j].pad & 1)) {
                if (fin.count(g) && fin[g] != b) {
                    printf("group %d finished by workgroups %d and %d\n", g, fin[g], b);
                    return 1;
                }
                fin[g] = b;
            }
            if (r[j].pad & 4) {
                merged = g;
            }
        }
        int cov = 0;
        for (int i = 0; i < PS_NW; i++) {  // streamer waves first, the control waves' shares at the end
            const int w = i < PS_NW - PS_NC ? i + PS_NC : i - (PS_NW - PS_NC);
            int       tb, te;
            ps_wave_range(T, w, cs3, tb, te);
            if (te > tb && tb != cov) {
                printf("workgroup %d wave %d: share starts at %d, expected %d\n", b, w, tb, cov);
                return 1;
            }
            cov = te > tb ? te : cov;
        }
        if (cov != T) {
            printf("workgroup %d: the waves cover %d of %d tiles\n", b, cov, T);
            return 1;
        }
        maxT = std::max(maxT, T);
        minT = std::min(minT, T);
        int       ent[64];
        const int c = ps_own_remote(b, NB, NG, KT_a, KT_b, TK, M, Il, cs3, tmp, ent, 64);
        maxrem      = std::max(maxrem, c);
        int pieces = 0;
        for (int b2 = 0; b2 < NB && merged >= 0; b2++) {
            if (b2 == b) {
                continue;
            }
            RunRec    r2[PS_RMAX];
            const int n2 = ps_own_runs(b2, NB, NG, KT_a, KT_b, TK, M, Il, r2);
            int       T2 = 0, pre = 0;
            for (int j = 0; j < n2; j++) {
                T2 += r2[j].nt;
            }
            for (int j = 0; j < n2; j++) {
                if (r2[j].grp == merged) {
                    for (int w = 0; w < PS_NW; w++) {
                        int tb, te;
                        ps_wave_range(T2, w, cs3, tb, te);
                        pieces += std::min(te, pre + r2[j].nt) > std::max(tb, pre) ? 1 : 0;
                    }
                }
                pre += r2[j].nt;
            }
        }
        if (pieces != c) {
            printf("workgroup %d: %d remote pieces of group %d, its list has %d\n", b, pieces, merged, c);
            return 1;
        }
    }
    for (int g = 0; g < NG; g++) {
        for (int t = 0; t < KT_b; t++) {
            if (covF[g][t] != 1) {
                printf("FFN2 group %d tile %d streamed %d times\n", g, t, covF[g][t]);
                return 1;
            }
        }
        for (int t = 0; t < KT_a; t++) {
            if (covO[g][t] != 1) {
                printf("out-proj group %d tile %d streamed %d times\n", g, t, covO[g][t]);
                return 1;
            }
        }
        if (!fin.count(g)) {
            printf("group %d is never finished\n", g);
            return 1;
        }
    }
    printf("NB %d NG %d KT %d / %d: ok, %d..%d tiles per workgroup, <= %d remote partials\n", NB, NG, KT_a, KT_b, minT, maxT, maxrem);
    return 0;
}

int main()
{
    int bad = 0;
    for (int tp : {1, 2, 4, 8}) {  // CodeFuse-13B int8 shards
        bad += check(256, 320, 80 / tp, 320 / tp, 64, 1, 20480 / tp, 10);
    }
    bad += check(256, 320, 160, 640, 32, 1, 20480, 10);  // fp16 weights
    bad += check(128, 320, 80, 320, 64, 2, 20480, 10);   // two processes on one GPU, two rows
    bad += check(32, 320, 10, 40, 64, 1, 2560, 10);      // a local group of eight ranks
    bad += check(32, 64, 16, 64, 64, 2, 4096, 10);       // the engine test's 1024-hidden model at 32 / 48 workgroups
    bad += check(48, 64, 16, 64, 64, 2, 4096, 10);
    bad += check(48, 64, 32, 128, 32, 2, 4096, 10);
    return bad;
}

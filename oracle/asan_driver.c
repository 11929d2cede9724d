/* oracle/asan_driver.c -- runs the C restatement under AddressSanitizer + UBSan (CPU only; GPU
 * sanitizers are not available on the pool).  Built and run by tests/test_oracle_sanitizers.py.
 * TEST INFRASTRUCTURE ONLY. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ssd_oracle.h"

static const char *HARVEST[] = {"@@@@@@@@", "@P A AP@", "@ AAA  @", "@PA  AP@", "@  A   @", "@P AA P@", "@@@@@@@@"};
static const char *CLEANUP[] = {"@@@@@@@@", "@HRP BB@", "@RH PBB@", "@HHSSBP@", "@RRP BB@", "@HR  PB@", "@@@@@@@@"};

static int run(int game, const char **rows, int H, int W, int E, int N) {
    char *map = (char *)malloc((size_t)H * W);
    for (int r = 0; r < H; ++r) memcpy(map + (size_t)r * W, rows[r], (size_t)W);
    uint8_t lut[128 * 3];
    for (int i = 0; i < 128 * 3; ++i) lut[i] = (uint8_t)(i * 7);
    ssd_oracle *o = ssd_oracle_create(game, H, W, map, E, N, 7, 5, 12345u, 3u, lut);
    if (!o) return 1;
    const int V = 15;
    uint8_t *obs = (uint8_t *)malloc((size_t)E * N * V * V * 3);
    int32_t *rew = (int32_t *)malloc(sizeof(int32_t) * E * N), *act = (int32_t *)malloc(sizeof(int32_t) * E * N);
    uint8_t *done = (uint8_t *)malloc((size_t)E * N), *mask = (uint8_t *)malloc((size_t)E);
    long total = 0;
    if (ssd_oracle_reset(o, NULL, obs)) return 2;
    for (int s = 0; s < 400; ++s) {
        if (ssd_oracle_step_random(o, game == 0 ? 8 : 9, act, obs, rew, done)) return 3;
        for (int i = 0; i < E * N; ++i) total += rew[i];
        if (s % 50 == 49) {
            for (int e = 0; e < E; ++e) mask[e] = (uint8_t)((e + s) & 1);
            if (ssd_oracle_reset(o, mask, obs)) return 4;
        }
    }
    if (ssd_oracle_observe(o, 0, obs)) return 5;
    printf("game %d ok, reward sum %ld\n", game, total);
    free(map); free(obs); free(rew); free(act); free(done); free(mask);
    ssd_oracle_destroy(o);
    return 0;
}

int main(void) {
    int rc = run(0, HARVEST, 7, 8, 16, 6);
    if (rc) return rc;
    return run(1, CLEANUP, 7, 8, 16, 5);
}

/* Sanitiser self-test of the oracle (CPU only; GPU AddressSanitizer is not available on the pool):
 * built with -fsanitize=address,undefined by `make -C oracle selftest` and run by tests/test_oracle_sanitized.py.
 * Renders a two-triangle quad through every entry point and checks the pixel count. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_fk(const double *, const double *, const double *, double *);
void orc_mvp(const double *, const double *, int, float *);
void orc_raster(const float *, const int32_t *, const int32_t *, const int32_t *, int, const float *, int, int, uint32_t *);
void orc_resolve(const uint32_t *, int, double, double, float *, uint8_t *);
void orc_sums(const uint32_t *, int, int, double, double, int, int, const uint64_t *, const float *, const int32_t *, uint64_t *);
double orc_finalize(const uint64_t *, int, int, double, const uint8_t *);
void orc_eval_batch(const float *, const int32_t *, const int32_t *, const int32_t *, const double *, const double *, const double *,
                    int, int, double, double, int, int, const uint64_t *, const float *, const int32_t *, const uint8_t *,
                    const double *, int, double *, uint64_t *, int);
void orc_coverage_batch(const float *, const int32_t *, const int32_t *, const int32_t *, const double *, const double *,
                        const double *, int, int, int, const double *, int, uint8_t *, int);
int orc_sum_words(void);

int main(void)
{
    enum { W = 64, H = 48 };
    const double n = 0.05, f = 100.0;
    double P[16] = {0};
    P[0] = 2.0 * 64 / W; P[5] = 2.0 * 64 / H; P[10] = (f + n) / (n - f); P[11] = 2 * f * n / (n - f); P[14] = -1;
    /* quad covering window x 24..40, y 16..40 at z = -2 (two CCW triangles), one link */
    float verts[12] = {-0.25f, -0.25f, -2, 0.25f, -0.25f, -2, 0.25f, 0.5f, -2, -0.25f, 0.5f, -2};
    int32_t faces[6] = {0, 1, 2, 0, 2, 3}, voff[2] = {0, 4}, toff[2] = {0, 2};
    double fixed[72] = {0}, axes[18] = {0}, q[12] = {0}, fk[84];
    for (int i = 0; i < 6; i++) { fixed[12 * i] = fixed[12 * i + 5] = fixed[12 * i + 10] = 1; axes[3 * i + 2] = 1; }
    float mvp[16];
    uint32_t *key = malloc(sizeof(uint32_t) * W * H);
    float *depth = malloc(sizeof(float) * W * H);
    uint8_t *ids = malloc(W * H), *cover = malloc(W * H), flags[8] = {0};
    uint64_t *tq = calloc(W * H, sizeof(uint64_t)), sums[64];
    orc_fk(fixed, axes, q, fk);
    orc_mvp(P, fk, 1, mvp);
    orc_raster(verts, faces, voff, toff, 1, mvp, W, H, key);
    orc_resolve(key, W * H, n, f, depth, ids);
    int covered = 0;
    for (int i = 0; i < W * H; i++) covered += ids[i] == 0;
    if (covered != 16 * 24) { printf("FAIL coverage %d\n", covered); return 1; }
    for (int loss = 0; loss < 4; loss++) {
        int32_t crop[4] = {4, 40, 10, 50};
        orc_sums(key, W, H, n, f, loss, 1, tq, depth, crop, sums);
        (void)orc_finalize(sums, loss, 1, (double)W * H, flags);
    }
    double err[2];
    uint64_t s2[2 * 64];
    if (orc_sum_words() > 64) return 2;
    orc_eval_batch(verts, faces, voff, toff, fixed, axes, P, W, H, n, f, 1, 1, tq, depth, NULL, flags, q, 2, err, s2, 2);
    orc_coverage_batch(verts, faces, voff, toff, fixed, axes, P, W, H, 1, q, 2, cover, 2);
    int c2 = 0;
    for (int i = 0; i < W * H; i++) c2 += cover[i];
    if (c2 != covered) { printf("FAIL coverage batch %d\n", c2); return 3; }
    free(key); free(depth); free(ids); free(cover); free(tq);
    printf("oracle selftest ok: %d samples, err %g\n", covered, err[0]);
    return 0;
}

// Measurement tool (CPU; driven by tools/dfm_fixed_points.py): iterates MS-DFM's update operator to a float
// fixed point in an evaluation order of choice, to show how far apart valid orders land.
//   op 0: the level-0 operator  min_rhs<0>  (DynamicFastMarching_impl.h:157-210): best cell of each pair, one
//         quadratic per stencil
//   op 1: the level-1 operator: the smallest of the eight candidates of min_rhs_decreased_neighbor (:270-313),
//         which is what DFMPlanner<1>::plan keeps in RHS (:79-86)
//   mode 0: Jacobi over the active set (an element rises only in sweeps of its colour of a 4-colouring)
//   mode 1: raster Gauss-Seidel sweeps in the four directions, replace semantics, until a round changes nothing
// args: N cost.bin goal_x goal_y op mode out.bin
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
static const float S2 = 1.41421356237309504880168872420969807856967187537694f;
static inline float Q(float a, float b, float th) {
    float ga = a < b ? a : b, gb = a < b ? b : a;
    if (ga == INFINITY && gb == INFINITY) return INFINITY;
    float d = gb - ga;
    if (th > d) return ((ga + gb) + sqrtf(2.0f * (th * th) - d * d)) * 0.5f;
    return ga + th;
}
static inline float mn(float a, float b) { return a < b ? a : b; }
static int N; static float *tau; static int gx, gy; static int OP;
#define AT(G, x, y) (((x) < 0 || (y) < 0 || (x) >= N || (y) >= N) ? INFINITY : (G)[(size_t)(x) * N + (y)])
static float evalF(const float *G, int x, int y) {
    if (x == gx && y == gy) return 0.0f;
    float t = tau[(size_t)x * N + y];
    if (t == INFINITY) return INFINITY;
    float T = AT(G, x - 1, y), B = AT(G, x + 1, y), L = AT(G, x, y - 1), R = AT(G, x, y + 1);
    float TL = AT(G, x - 1, y - 1), BR = AT(G, x + 1, y + 1), BL = AT(G, x + 1, y - 1), TR = AT(G, x - 1, y + 1);
    float th2 = t * S2;
    if (OP == 0) {   // level 0: F
        float o = Q(T < B ? T : B, L < R ? L : R, t), d = Q(TL < BR ? TL : BR, BL < TR ? BL : TR, th2);
        return d < o ? d : o;
    }
    float lr = L < R ? L : R, tb = T < B ? T : B, d1 = TL < BR ? TL : BR, d2 = BL < TR ? BL : TR;
    float r = Q(T, lr, t);
    r = mn(r, Q(B, lr, t)); r = mn(r, Q(L, tb, t)); r = mn(r, Q(R, tb, t));
    r = mn(r, Q(TR, d1, th2)); r = mn(r, Q(BL, d1, th2)); r = mn(r, Q(TL, d2, th2)); r = mn(r, Q(BR, d2, th2));
    return r;
}
int main(int argc, char **argv) {
    N = atoi(argv[1]); const char *costf = argv[2]; gx = atoi(argv[3]); gy = atoi(argv[4]); OP = atoi(argv[5]);
    int mode = atoi(argv[6]); const char *outf = argv[7];
    size_t n = (size_t)N * N;
    uint8_t *c = malloc(n); FILE *f = fopen(costf, "rb"); if (fread(c, 1, n, f) != n) return 1; fclose(f);
    tau = malloc(n * 4); for (size_t i = 0; i < n; ++i) tau[i] = c[i] >= 255 ? INFINITY : (float)c[i];
    float *G = malloc(n * 4), *H = malloc(n * 4);
    uint8_t *act = calloc(n, 1), *nact = calloc(n, 1);
    for (size_t i = 0; i < n; ++i) G[i] = INFINITY;
    G[(size_t)gx * N + gy] = 0; memcpy(H, G, n * 4);
    for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) { int x = gx + dx, y = gy + dy; if (x >= 0 && y >= 0 && x < N && y < N) act[(size_t)x * N + y] = 1; }
    long it = 0, total = 0;
    if (mode == 0) {   // Jacobi over the active set, rises only in sweeps of the element's colour
        for (;; ++it) {
            long ch = 0, pend = 0;
            #pragma omp parallel for schedule(dynamic, 16) reduction(+:ch, pend)
            for (int x = 0; x < N; ++x) for (int y = 0; y < N; ++y) {
                size_t i = (size_t)x * N + y;
                if (!act[i]) continue;
                float r = evalF(G, x, y), g = G[i];
                int col = (x & 1) | ((y & 1) << 1);
                if (r != g) { if (r < g || col == (it & 3)) { H[i] = r; ++ch; } else { ++pend; nact[i] = 1; } }
            }
            if (!ch && !pend) break;
            #pragma omp parallel for schedule(dynamic, 16)
            for (int x = 0; x < N; ++x) for (int y = 0; y < N; ++y) {
                size_t i = (size_t)x * N + y;
                if (H[i] != G[i]) { G[i] = H[i]; for (int dx = -1; dx <= 1; ++dx) for (int dy = -1; dy <= 1; ++dy) { int a = x + dx, b = y + dy; if (a >= 0 && b >= 0 && a < N && b < N) nact[(size_t)a * N + b] = 1; } }
            }
            uint8_t *t = act; act = nact; nact = t; memset(nact, 0, n);
            total += ch;
            if (it % 500 == 0) fprintf(stderr, "it %ld changes %ld pending %ld\n", it, ch, pend);
        }
    } else {           // Gauss-Seidel raster sweeps in four directions, replace semantics, until a full round changes nothing
        for (;; ++it) {
            long ch = 0;
            for (int dir = 0; dir < 4; ++dir)
                for (int a = 0; a < N; ++a) for (int b = 0; b < N; ++b) {
                    int x = (dir & 1) ? N - 1 - a : a, y = (dir & 2) ? N - 1 - b : b;
                    size_t i = (size_t)x * N + y;
                    float r = evalF(G, x, y);
                    if (r != G[i]) { G[i] = r; ++ch; }
                }
            total += ch;
            fprintf(stderr, "round %ld changes %ld\n", it, ch);
            if (!ch || it > 400) break;
        }
    }
    fprintf(stderr, "iterations %ld, total changes %ld\n", it, total);
    f = fopen(outf, "wb"); fwrite(G, 4, n, f); fclose(f);
    return 0;
}

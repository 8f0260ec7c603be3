/* tools/percall/percall_bench.c — BASELINE config 5 the way SURVEY.md §8(d) literally describes it: "written element by element".
 * 800 000 calls of dsa_mat_set (setindex!(m, val, row, col), reference src/matrix.jl:43-62) through the C ABI, one call per element, the way
 * a Julia host's `A[i, j] = v` arrives (ccall per element); the library queues the writes on the host and applies them in order behind the
 * next observing call or at 65 536 pending writes (include/dsa.h: write combining).  A column-generation host observes the matrix after
 * every batch of columns: here a dsa_mat_nnz every `every` columns.  Prints one JSON object.  Plain C against include/dsa.h: this is
 * also the smallest complete client of the boundary.
 *   usage: percall_bench <rows> <columns> <rows per column> <observe every N columns> <repeats>
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include "dsa.h"

static uint64_t sm_state;
static uint64_t splitmix(void) {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int cmp_i64(const void* a, const void* b) { const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b; return x < y ? -1 : x > y; }
#define CHK(e) do { int32_t rc_ = (e); if (rc_ != DSA_OK) { fprintf(stderr, "%s: status %d: %s\n", #e, (int)rc_, dsa_last_error_message()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int64_t m = argc > 1 ? atoll(argv[1]) : 100000, ncols = argc > 2 ? atoll(argv[2]) : 50000, per = argc > 3 ? atoll(argv[3]) : 16;
    const int64_t every = argc > 4 ? atoll(argv[4]) : 1000;
    const int reps = argc > 5 ? atoi(argv[5]) : 3;
    const int64_t n = ncols * per;
    int64_t* I = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    double* V = (double*)malloc((size_t)n * sizeof(double));
    if (!I || !V) return 2;
    /* per distinct rows per column, ascending inside a column (own generator: the parity of this loop is the suite's business, this is a timing leg) */
    sm_state = 11;
    for (int64_t c = 0; c < ncols; ++c) {
        int64_t* r = I + c * per;
        for (int64_t k = 0; k < per; ++k) {
            int dup;
            do { r[k] = 1 + (int64_t)(splitmix() % (uint64_t)m); dup = 0; for (int64_t q = 0; q < k; ++q) dup |= r[q] == r[k]; } while (dup);
        }
        qsort(r, (size_t)per, sizeof(int64_t), cmp_i64);
    }
    sm_state = 12;
    for (int64_t k = 0; k < n; ++k) V[k] = 1.0 + (double)(splitmix() >> 11) * (1.0 / 9007199254740992.0);
    double best = 1e30, sum = 0.0, call_s_min = 1e30, times[16];
    int64_t nnz = 0;
    for (int rep = 0; rep < reps + 1; ++rep) {            /* the first repeat is the warm-up (code objects, pools, graphs) */
        dsa_mat_t* A = NULL;
        CHK(dsa_mat_create_empty(0, &A));
        const double t0 = now_s();
        double in_calls = 0.0;
        for (int64_t c = 0; c < ncols; ++c) {
            const double tc = now_s();
            for (int64_t k = 0; k < per; ++k) CHK(dsa_mat_set(A, V[c * per + k], I[c * per + k], c + 1));
            in_calls += now_s() - tc;
            if ((c + 1) % every == 0) CHK(dsa_mat_nnz(A, &nnz));      /* the host looks at the matrix: queued writes are applied */
        }
        CHK(dsa_mat_nnz(A, &nnz));
        CHK(dsa_mat_sync(A));
        const double dt = now_s() - t0;
        if (rep > 0) { times[rep - 1] = dt; sum += dt; if (dt < best) best = dt; if (in_calls < call_s_min) call_s_min = in_calls; }
        CHK(dsa_mat_destroy(A));
    }
    /* median of the timed repeats */
    for (int a = 0; a < reps; ++a) for (int b = a + 1; b < reps; ++b) if (times[b] < times[a]) { const double t = times[a]; times[a] = times[b]; times[b] = t; }
    const double med = times[reps / 2];
    printf("{\"calls\": %lld, \"columns\": %lld, \"observe_every_columns\": %lld, \"repeats\": %d, \"median_s\": %.4f, \"best_s\": %.4f, "
           "\"calls_per_s\": %.1f, \"columns_per_s\": %.1f, \"ns_per_call_inside_dsa_mat_set\": %.1f, \"nnz\": %lld}\n",
           (long long)n, (long long)ncols, (long long)every, reps, med, best, (double)n / med, (double)ncols / med, 1e9 * call_s_min / (double)n, (long long)nnz);
    free(I); free(V);
    return 0;
}

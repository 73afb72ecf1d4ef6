/* ThreadSanitizer driver of libhalo_host's writer functions (tests/test_sanitizers.py): N threads retire images concurrently --
 * the lazily built CRC / Huffman tables, the per-thread scratch with its exit-time destructor and the file writes are what could
 * race.  Built TOGETHER with halo_amd/csrc/halo_host.c under -fsanitize=thread; exit code 0 and no report = clean.
 *   gcc -O1 -g -fsanitize=thread -pthread halo_host.c host_tsan.c -o host_tsan && ./host_tsan <dir> [threads] [images per thread] */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/halo_host.h"

enum { H = 96, W = 160, NPICK = 40 };
static const char *dir;
static int per_thread = 6;

static void *worker(void *arg)
{
    const long id = (long)arg;
    uint64_t s = 88172645463325252ull + (uint64_t)id * 7919u;
    int64_t *gt = malloc(sizeof(int64_t) * H * W), *om = malloc(sizeof(int64_t) * H * W);
    uint8_t *act = malloc(H * W), *sel = malloc(H * W), *mask = malloc(H * W), *png = malloc(halo_png_gray8_bound(H, W));
    double picks[NPICK * 3];
    long bad = 0;
    for (int it = 0; it < per_thread; ++it) {
        for (int i = 0; i < H * W; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            gt[i] = (int64_t)(s % 19); om[i] = 255; act[i] = (s >> 20) % 50 == 0; sel[i] = 0;
        }
        for (int p = 0; p < NPICK; ++p) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            picks[3 * p] = (double)(s % H); picks[3 * p + 1] = (double)((s >> 16) % W); picks[3 * p + 2] = 1.0 - p * 0.01;
        }
        char p1[512], p2[512];
        snprintf(p1, sizeof p1, "%s/t%ld_%d.png", dir, id, it);
        /* the one-call writer without an indicator template, then the pieces on their own */
        if (halo_retire_image(p1, 0, om, 8, gt, 8, H, W, picks, NPICK, 1, act, sel, 5, 0, 0, 0, 0, 0, 0) != 0) ++bad;
        if (halo_compose_mask(mask, om, 8, gt, 8, H, W, picks, NPICK, 1) != 0) ++bad;
        if (halo_png_gray8_encode(mask, H, W, W, png, halo_png_gray8_bound(H, W)) == 0) ++bad;
        snprintf(p2, sizeof p2, "%s/t%ld_%d_b.png", dir, id, it);
        if (halo_png_gray8_write(p2, mask, H, W, W) != 0) ++bad;
        if (halo_compose_indicators(act, sel, act, sel, H, W, picks, NPICK, 1, 5) != 0) ++bad;
        if (halo_crc32(0, mask, H * W) == 0) ++bad;
        /* both files of an image must hold the same bytes: the retire call and the pieces compute the same mask */
        FILE *f1 = fopen(p1, "rb"), *f2 = fopen(p2, "rb");
        if (!f1 || !f2) ++bad;
        else {
            int c1, c2;
            do { c1 = fgetc(f1); c2 = fgetc(f2); if (c1 != c2) { ++bad; break; } } while (c1 != EOF);
        }
        if (f1) fclose(f1);
        if (f2) fclose(f2);
        remove(p1); remove(p2);
    }
    if (id % 2) halo_host_thread_release();            /* half of the threads release explicitly, the others at exit */
    free(gt); free(om); free(act); free(sel); free(mask); free(png);
    return (void *)bad;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    dir = argv[1];
    const int n = argc > 2 ? atoi(argv[2]) : 12;
    if (argc > 3) per_thread = atoi(argv[3]);
    long bad = 0;
    for (int round = 0; round < 2; ++round) {           /* two generations of threads: scratch of exited threads is gone, tables stay */
        pthread_t th[64];
        for (long i = 0; i < n; ++i) pthread_create(&th[i], 0, worker, (void *)i);
        for (int i = 0; i < n; ++i) { void *r; pthread_join(th[i], &r); bad += (long)r; }
    }
    printf("%d threads x %d images x 2 generations: %ld failed calls\n", n, per_thread, bad);
    return bad != 0;
}

/* examples/extract_match.c -- the C ABI of libygzf from plain C99 (no C++, no Python): extract ORB features of two frames that are
 * resident on the host, match frame 1 against frame 0 with ORBmatcher::SearchByProjection(Cur, Last) semantics, print the counts.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/extract_match.c -Lorb_ygz_slam_amd/lib -lygzf -Wl,-rpath,$PWD/orb_ygz_slam_amd/lib -o extract_match
 *   ./extract_match            (needs an MI355X; there is no CPU fallback)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ygzf.h"

static unsigned lcg(unsigned *s) { return *s = *s * 1664525u + 1013904223u; }

/* a frame with structure: random axis-aligned rectangles on a noisy background, shifted by (dx, dy) */
static void render(uint8_t *img, int w, int h, int dx, int dy) {
    unsigned s = 12345u;
    int x, y, k;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) img[y * w + x] = (uint8_t) (100 + (lcg(&s) >> 24) % 6);
    s = 999u;
    for (k = 0; k < 220; k++) {
        const int rx = (int) (lcg(&s) % (unsigned) w) + dx, ry = (int) (lcg(&s) % (unsigned) h) + dy;
        const int rw = 8 + (int) (lcg(&s) % 40u), rh = 8 + (int) (lcg(&s) % 40u);
        const uint8_t v = (uint8_t) (lcg(&s) >> 24);
        for (y = ry; y < ry + rh; y++)
            for (x = rx; x < rx + rw; x++)
                if (x >= 0 && y >= 0 && x < w && y < h) img[y * w + x] = v;
    }
}

int main(void) {
    const int w = 752, h = 480;
    ygzf_extractor_cfg cfg = {1000, 1.2f, 8, 20, 7};
    ygzf_camera cam = {458.654f, 457.296f, 367.215f, 248.375f, 0.f, 0.f, 0.f, 0.f, 752.f, 480.f};
    ygzf_ctx *ctx = NULL;
    uint8_t *frames = (uint8_t *) malloc((size_t) 2 * w * h);
    int counts[2], nmatch[2], rc;
    if (ygzf_create(0, &cfg, w, h, 2, &ctx) != YGZF_OK) {
        fprintf(stderr, "ygzf_create: %s\n", ygzf_last_error(NULL));
        return 1;
    }
    render(frames, w, h, 0, 0);
    render(frames + (size_t) w * h, w, h, 3, 2);
    rc = ygzf_extract_batch_host(ctx, frames, 2, w, h, w, (size_t) w * h);
    if (rc == YGZF_OK) rc = ygzf_match_batch_prev(ctx, &cam, 15.f, 1, 1, 1);   /* frame f against frame f-1, th = 15, mono */
    if (rc == YGZF_OK) rc = ygzf_batch_counts(ctx, counts);
    if (rc == YGZF_OK) rc = ygzf_match_counts(ctx, nmatch);
    if (rc != YGZF_OK) {
        fprintf(stderr, "libygzf: %s\n", ygzf_last_error(ctx));
        return 1;
    }
    printf("keypoints: %d %d   matches of frame 1 against frame 0: %d\n", counts[0], counts[1], nmatch[1]);
    ygzf_destroy(ctx);
    free(frames);
    return counts[0] > 100 && counts[1] > 100 && nmatch[1] > 50 ? 0 : 2;
}

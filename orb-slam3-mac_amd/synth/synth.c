/*
 * synth.c -- seeded synthetic image generator for the ORB bench / parity tests
 * (SURVEY.md section 8d, config #2/#3).  Host-only C, no dependencies.  NOT part of the
 * oracle and NOT part of the measured path: it only manufactures input bytes so that
 * the CPU oracle and the HIP path see identical images.
 *
 * Scene model: mid-grey background + N random shapes (axis-aligned rects, rotated
 * rects, discs; uniform intensity; 6..60 px) drawn painter's-order on a canvas larger
 * than the frame, then additive noise (sum of 4 uniforms, sigma ~3.4).  Frames come in
 * sequences of SYNTH_SEQ_LEN: frame j of a sequence is the sequence's scene rotated by
 * j*dtheta (|dtheta| <= 5 deg) about the image centre and translated by j*(dx,dy)
 * (|dx|,|dy| <= 8 px), so consecutive frames have a known rigid relation.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>

#define SYNTH_SEQ_LEN 8

typedef struct { uint64_t state, inc; } pcg32_t;

static uint32_t pcg32_next(pcg32_t *r)
{
    uint64_t old = r->state;
    r->state = old * 6364136223846793005ULL + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((-rot) & 31));
}
static void pcg32_seed(pcg32_t *r, uint64_t seed, uint64_t seq)
{
    r->state = 0; r->inc = (seq << 1u) | 1u;
    pcg32_next(r); r->state += seed; pcg32_next(r);
}
static float pcg32_unit(pcg32_t *r) { return (float)(pcg32_next(r) >> 8) * (1.0f / 16777216.0f); }

void synth_frame_transform(uint64_t seed, int frame_id, float *theta_rad, float *tx, float *ty)
{
    pcg32_t r;
    int seq = frame_id / SYNTH_SEQ_LEN, j = frame_id % SYNTH_SEQ_LEN;
    pcg32_seed(&r, seed ^ 0x9e3779b97f4a7c15ULL, (uint64_t)seq * 2 + 1);
    float dth = (pcg32_unit(&r) * 2.f - 1.f) * (5.0f * 3.14159265f / 180.f);
    float dx = (pcg32_unit(&r) * 2.f - 1.f) * 8.f;
    float dy = (pcg32_unit(&r) * 2.f - 1.f) * 8.f;
    *theta_rad = j * dth; *tx = j * dx; *ty = j * dy;
}

void synth_frame(uint8_t *out, int w, int h, int stride, uint64_t seed, int frame_id)
{
    pcg32_t rs, rn;
    int seq = frame_id / SYNTH_SEQ_LEN;
    float th, tx, ty;
    synth_frame_transform(seed, frame_id, &th, &tx, &ty);
    pcg32_seed(&rs, seed, (uint64_t)seq * 2);
    for (int y = 0; y < h; y++) memset(out + (size_t)y * stride, 128, w);
    const float cxi = 0.5f * w, cyi = 0.5f * h;
    const float ct = cosf(th), st = sinf(th);
    const float margin = 0.25f * (w > h ? w : h);
    int nshapes = (int)(400.0 * ((double)w * h) / (640.0 * 480.0) * 2.25);   /* canvas is 1.5x per side */
    for (int s = 0; s < nshapes; s++) {
        int type = pcg32_next(&rs) % 3;
        float sx = pcg32_unit(&rs) * (w + 2 * margin) - margin - cxi;   /* scene coords, centre origin */
        float sy = pcg32_unit(&rs) * (h + 2 * margin) - margin - cyi;
        float a = 3.f + pcg32_unit(&rs) * 27.f, b = 3.f + pcg32_unit(&rs) * 27.f;
        float phi = type == 1 ? pcg32_unit(&rs) * 3.14159265f : 0.f;
        int val = pcg32_next(&rs) & 255;
        /* frame coords of the shape centre and its total rotation */
        float fx = ct * sx - st * sy + tx + cxi, fy = st * sx + ct * sy + ty + cyi;
        float rot = phi + th;
        float cr = cosf(rot), sr = sinf(rot);
        float rad = type == 2 ? a : sqrtf(a * a + b * b);
        int x0 = (int)floorf(fx - rad), x1 = (int)ceilf(fx + rad);
        int y0 = (int)floorf(fy - rad), y1 = (int)ceilf(fy + rad);
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0;
        if (x1 > w - 1) x1 = w - 1; if (y1 > h - 1) y1 = h - 1;
        for (int y = y0; y <= y1; y++) {
            uint8_t *row = out + (size_t)y * stride;
            for (int x = x0; x <= x1; x++) {
                float px = x - fx, py = y - fy;
                int in;
                if (type == 2) in = px * px + py * py <= a * a;
                else {
                    float qx = cr * px + sr * py, qy = -sr * px + cr * py;
                    in = fabsf(qx) <= a && fabsf(qy) <= b;
                }
                if (in) row[x] = (uint8_t)val;
            }
        }
    }
    pcg32_seed(&rn, seed ^ 0xda3e39cb94b95bdbULL, (uint64_t)frame_id);
    for (int y = 0; y < h; y++) {
        uint8_t *row = out + (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            uint32_t r = pcg32_next(&rn);
            int nz = (int)((r & 255) % 6 + ((r >> 8) & 255) % 6 + ((r >> 16) & 255) % 6 + ((r >> 24) & 255) % 6) - 10;
            int v = row[x] + nz;
            row[x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

void synth_batch(uint8_t *out, int w, int h, int n, uint64_t seed, int first_frame)
{
    for (int i = 0; i < n; i++)
        synth_frame(out + (size_t)i * w * h, w, h, w, seed, first_frame + i);
}

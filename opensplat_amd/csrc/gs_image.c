/* gs_image.c — JPEG decoder (baseline, extended sequential and progressive Huffman) of
 * libgsplat_image.so (include/gsplat_image.h).
 *
 * Written from the JPEG standard (ITU T.81) and the documented behaviour of the IJG library's
 * default decoding path, which is what cv::imread runs for OpenSplat's training images
 * (cv_utils.cpp:3-14): sequential Huffman decoding, de-quantisation, the "islow" integer inverse
 * DCT (13-bit constants, two passes), "fancy" chroma upsampling (triangle filter; h2v1 and h2v2, any
 * other integral ratio by replication) and fixed-point YCbCr -> RGB (16-bit tables).  Those four
 * steps determine the pixel values bit for bit; tests/test_image.py pins the result against libjpeg
 * itself (through Pillow) on 4:4:4 / 4:2:2 / 4:2:0 / greyscale files of ragged sizes, several
 * qualities, optimised Huffman tables and restart markers.
 *
 * A single-scan sequential file (what cameras write) is decoded MCU by MCU straight into the sample
 * planes.  Progressive files (SOF2: spectral selection and successive approximation, T.81 annex G) and
 * sequential files with several scans collect their coefficients for the whole image first, scan by
 * scan, and are transformed at the end — libjpeg's final output for a complete file.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/gsplat_image.h"

#define MAX_COMPS 3

typedef struct {
    uint8_t bits[17];      /* bits[k] = number of codes of length k */
    uint8_t vals[256];
    int maxcode[18];       /* largest code of length k (-1 if none) */
    int valptr[17];        /* index of the first value of length k */
    int mincode[17];
    /* 9-bit look-ahead table: (length << 8) | value, 0 = longer than 9 bits */
    uint16_t look[512];
    int present;
} Huff;

typedef struct {
    int id, h, v, tq, td, ta;
    int blocks_w, blocks_h;    /* padded size in blocks (whole MCUs) */
    int ds_w, ds_h;            /* down-sampled size in samples: ceil(W * h / hmax) ... */
    uint8_t *plane;            /* blocks_w * 8 x blocks_h * 8 samples */
    int pred;
} Comp;

typedef struct {
    const uint8_t *p, *end;
    uint32_t bitbuf;
    int bitcnt;
    int hit_marker;            /* marker byte met inside entropy data (0 = none) */
} Bits;

typedef struct {
    int W, H, ncomp, hmax, vmax;
    Comp c[MAX_COMPS];
    uint16_t qt[4][64];
    int qt_present[4];
    Huff dc[4], ac[4];
    int restart_interval;
    int adobe_transform;       /* -1 = no Adobe marker */
    int progressive;
    int have_sof;
    /* the scan header last parsed (SOS) */
    int ns, scomp[MAX_COMPS];  /* components of the scan: indices into c[] */
    int ss, se, ah, al;
} Jpeg;

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

const char *gs_image_strerror(int status) {
    switch (status) {
    case GS_IMG_OK: return "ok";
    case GS_IMG_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GS_IMG_ERR_UNSUPPORTED: return "unsupported JPEG variant (arithmetic-coded, lossless, 12-bit, CMYK)";
    case GS_IMG_ERR_CORRUPT: return "corrupt or truncated JPEG stream";
    default: return "unknown status";
    }
}

/* ---- Huffman tables (T.81 annex C / F.2.2.3) ------------------------------------------------- */
static int build_huff(Huff *h) {
    int code = 0, k = 0;
    int huffsize[257], huffcode[257];
    int n = 0;
    for (int l = 1; l <= 16; l++)
        for (int i = 0; i < h->bits[l]; i++) huffsize[n++] = l;
    if (n > 256) return GS_IMG_ERR_CORRUPT;
    huffsize[n] = 0;
    int si = huffsize[0];
    while (k < n) {
        while (k < n && huffsize[k] == si) huffcode[k++] = code++;
        if (code > (1 << si)) return GS_IMG_ERR_CORRUPT;
        code <<= 1;
        si++;
    }
    int p = 0;
    for (int l = 1; l <= 16; l++) {
        if (h->bits[l]) {
            h->valptr[l] = p;
            h->mincode[l] = huffcode[p];
            p += h->bits[l];
            h->maxcode[l] = huffcode[p - 1];
        } else {
            h->maxcode[l] = -1;
            h->valptr[l] = 0;
            h->mincode[l] = 0;
        }
    }
    h->maxcode[17] = 0x7fffffff;
    memset(h->look, 0, sizeof(h->look));
    p = 0;
    for (int l = 1; l <= 9; l++)
        for (int i = 0; i < h->bits[l]; i++, p++) {
            const int first = huffcode[p] << (9 - l);
            for (int j = 0; j < (1 << (9 - l)); j++)
                h->look[first + j] = (uint16_t)((l << 8) | h->vals[p]);
        }
    h->present = 1;
    return GS_IMG_OK;
}

/* ---- bit reader (byte stuffing, markers) -------------------------------------------------------- */
static void fill_bits(Bits *b) {
    while (b->bitcnt <= 24) {
        int byte = 0;
        if (!b->hit_marker && b->p < b->end) {
            byte = *b->p++;
            if (byte == 0xFF) {
                int nxt = b->p < b->end ? *b->p : 0xD9;
                if (nxt == 0x00) {
                    b->p++;                      /* stuffed zero */
                } else {
                    b->hit_marker = nxt;         /* leave the marker for the caller, feed zeros */
                    b->p--;
                    byte = 0;
                }
            }
        }
        b->bitbuf |= (uint32_t)byte << (24 - b->bitcnt);
        b->bitcnt += 8;
    }
}
static inline int peek_bits(Bits *b, int n) { return (int)(b->bitbuf >> (32 - n)); }
static inline void drop_bits(Bits *b, int n) {
    b->bitbuf <<= n;
    b->bitcnt -= n;
}
static inline int get_bits(Bits *b, int n) {
    if (n == 0) return 0;
    if (b->bitcnt < n) fill_bits(b);
    const int v = peek_bits(b, n);
    drop_bits(b, n);
    return v;
}
static inline int decode_symbol(Bits *b, const Huff *h) {
    if (b->bitcnt < 16) fill_bits(b);
    const int look = h->look[peek_bits(b, 9)];
    if (look) {
        drop_bits(b, look >> 8);
        return look & 0xFF;
    }
    int code = peek_bits(b, 9), l = 9;
    drop_bits(b, 9);
    for (;;) {
        l++;
        if (l > 16) return -1;
        code = (code << 1) | get_bits(b, 1);
        if (code <= h->maxcode[l]) break;
    }
    return h->vals[h->valptr[l] + code - h->mincode[l]];
}
/* T.81 F.2.2.1 EXTEND */
static inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

/* ---- inverse DCT: the "islow" algorithm (13-bit fixed point, two passes) -------------------------- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1L << ((n)-1))) >> (n))

static inline uint8_t clamp_sample(long v) {
    v += 128;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

static void idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
    long ws[64];
    for (int c = 0; c < 8; c++) {
        const long in0 = (long)coef[c] * q[c], in1 = (long)coef[8 + c] * q[8 + c];
        const long in2 = (long)coef[16 + c] * q[16 + c], in3 = (long)coef[24 + c] * q[24 + c];
        const long in4 = (long)coef[32 + c] * q[32 + c], in5 = (long)coef[40 + c] * q[40 + c];
        const long in6 = (long)coef[48 + c] * q[48 + c], in7 = (long)coef[56 + c] * q[56 + c];
        long z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
        z2 = in2; z3 = in6;
        z1 = (z2 + z3) * FIX_0_541196100;
        tmp2 = z1 + z3 * (-FIX_1_847759065);
        tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in0; z3 = in4;
        tmp0 = (z2 + z3) * (1L << CONST_BITS);   /* (a shift of a negative value is undefined in C) */
        tmp1 = (z2 - z3) * (1L << CONST_BITS);
        tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
        tmp0 = in7; tmp1 = in5; tmp2 = in3; tmp3 = in1;
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
        z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        ws[c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        ws[56 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        ws[8 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        ws[48 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        ws[16 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        ws[40 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        ws[24 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        ws[32 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; r++) {
        const long *w = ws + 8 * r;
        uint8_t *o = out + (size_t)r * stride;
        long z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
        z2 = w[2]; z3 = w[6];
        z1 = (z2 + z3) * FIX_0_541196100;
        tmp2 = z1 + z3 * (-FIX_1_847759065);
        tmp3 = z1 + z2 * FIX_0_765366865;
        tmp0 = (w[0] + w[4]) * (1L << CONST_BITS);
        tmp1 = (w[0] - w[4]) * (1L << CONST_BITS);
        tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
        z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int sh = CONST_BITS + PASS1_BITS + 3;
        o[0] = clamp_sample(DESCALE(tmp10 + tmp3, sh));
        o[7] = clamp_sample(DESCALE(tmp10 - tmp3, sh));
        o[1] = clamp_sample(DESCALE(tmp11 + tmp2, sh));
        o[6] = clamp_sample(DESCALE(tmp11 - tmp2, sh));
        o[2] = clamp_sample(DESCALE(tmp12 + tmp1, sh));
        o[5] = clamp_sample(DESCALE(tmp12 - tmp1, sh));
        o[3] = clamp_sample(DESCALE(tmp13 + tmp0, sh));
        o[4] = clamp_sample(DESCALE(tmp13 - tmp0, sh));
    }
}

/* ---- header parsing ------------------------------------------------------------------------------- */
static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

/* Parses markers from *pp up to and including the next SOS: tables are updated, the scan header is
 * stored in j, *pp is left at the entropy-coded data.  Returns GS_IMG_OK (a scan follows), 1 (EOI or the
 * end of the data: no further scan) or an error.  Call with a zeroed-and-initialised j the first time
 * (jpeg_begin). */
static int jpeg_begin(Jpeg *j, const uint8_t *data, size_t size, const uint8_t **pp) {
    memset(j, 0, sizeof(*j));
    j->adobe_transform = -1;
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return GS_IMG_ERR_CORRUPT;
    *pp = data + 2;
    return GS_IMG_OK;
}
static int next_scan(Jpeg *j, const uint8_t **pp, const uint8_t *end) {
    const uint8_t *p = *pp;
    for (;;) {
        while (p < end && *p != 0xFF) p++;                 /* resynchronise */
        while (p < end && *p == 0xFF) p++;                 /* fill bytes */
        if (p >= end) return j->have_sof ? 1 : GS_IMG_ERR_CORRUPT;
        const int m = *p++;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01 || m == 0x00) continue;
        if (m == 0xD9) return 1;                           /* EOI */
        if (p + 2 > end) return GS_IMG_ERR_CORRUPT;
        const int len = rd16(p);
        if (len < 2 || p + len > end) return GS_IMG_ERR_CORRUPT;
        const uint8_t *s = p + 2, *se = p + len;
        switch (m) {
        case 0xC0: case 0xC1: case 0xC2: {                 /* baseline / extended sequential / progressive, Huffman */
            if (j->have_sof) return GS_IMG_ERR_CORRUPT;
            if (se - s < 6) return GS_IMG_ERR_CORRUPT;
            if (s[0] != 8) return GS_IMG_ERR_UNSUPPORTED;
            j->progressive = m == 0xC2;
            j->H = rd16(s + 1);
            j->W = rd16(s + 3);
            j->ncomp = s[5];
            if (j->W <= 0 || j->H <= 0) return GS_IMG_ERR_UNSUPPORTED;   /* (DNL-defined height) */
            if (j->ncomp != 1 && j->ncomp != 3) return GS_IMG_ERR_UNSUPPORTED;
            if (se - s < 6 + 3 * j->ncomp) return GS_IMG_ERR_CORRUPT;
            for (int i = 0; i < j->ncomp; i++) {
                Comp *c = &j->c[i];
                c->id = s[6 + 3 * i];
                c->h = s[7 + 3 * i] >> 4;
                c->v = s[7 + 3 * i] & 15;
                c->tq = s[8 + 3 * i];
                if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4 || c->tq > 3) return GS_IMG_ERR_CORRUPT;
                if (c->h > j->hmax) j->hmax = c->h;
                if (c->v > j->vmax) j->vmax = c->v;
            }
            j->have_sof = 1;
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB:
        case 0xCD: case 0xCE: case 0xCF:
            return GS_IMG_ERR_UNSUPPORTED;                 /* lossless, hierarchical, arithmetic */
        case 0xC4:                                         /* DHT */
            while (s < se) {
                if (se - s < 17) return GS_IMG_ERR_CORRUPT;
                const int tc = s[0] >> 4, th = s[0] & 15;
                if (tc > 1 || th > 3) return GS_IMG_ERR_CORRUPT;
                Huff *h = tc ? &j->ac[th] : &j->dc[th];
                int n = 0;
                h->bits[0] = 0;
                for (int k = 1; k <= 16; k++) n += (h->bits[k] = s[k]);
                if (n > 256 || se - s < 17 + n) return GS_IMG_ERR_CORRUPT;
                memcpy(h->vals, s + 17, (size_t)n);
                const int rc = build_huff(h);
                if (rc) return rc;
                s += 17 + n;
            }
            break;
        case 0xDB:                                         /* DQT */
            while (s < se) {
                const int pq = s[0] >> 4, tq = s[0] & 15;
                if (tq > 3 || pq > 1) return GS_IMG_ERR_CORRUPT;
                if (se - s < 1 + 64 * (pq + 1)) return GS_IMG_ERR_CORRUPT;
                for (int k = 0; k < 64; k++)
                    j->qt[tq][kZigzag[k]] = (uint16_t)(pq ? rd16(s + 1 + 2 * k) : s[1 + k]);
                j->qt_present[tq] = 1;
                s += 1 + 64 * (pq + 1);
            }
            break;
        case 0xDD:                                         /* DRI */
            if (se - s < 2) return GS_IMG_ERR_CORRUPT;
            j->restart_interval = rd16(s);
            break;
        case 0xEE:                                         /* APP14 "Adobe" */
            if (se - s >= 12 && memcmp(s, "Adobe", 5) == 0) j->adobe_transform = s[11];
            break;
        case 0xDA: {                                       /* SOS */
            if (!j->have_sof) return GS_IMG_ERR_CORRUPT;
            if (se - s < 1) return GS_IMG_ERR_CORRUPT;
            j->ns = s[0];
            if (j->ns < 1 || j->ns > j->ncomp || se - s < 1 + 2 * j->ns + 3) return GS_IMG_ERR_CORRUPT;
            for (int i = 0; i < j->ns; i++) {
                const int cid = s[1 + 2 * i], t = s[2 + 2 * i];
                int idx = -1;
                for (int k = 0; k < j->ncomp; k++)
                    if (j->c[k].id == cid) idx = k;
                /* (components of a scan follow the frame's order, T.81 B.2.3) */
                if (idx < 0 || (i > 0 && idx <= j->scomp[i - 1])) return GS_IMG_ERR_CORRUPT;
                j->scomp[i] = idx;
                j->c[idx].td = t >> 4;
                j->c[idx].ta = t & 15;
                if (j->c[idx].td > 3 || j->c[idx].ta > 3) return GS_IMG_ERR_CORRUPT;
            }
            const uint8_t *q = s + 1 + 2 * j->ns;
            j->ss = q[0];
            j->se = q[1];
            j->ah = q[2] >> 4;
            j->al = q[2] & 15;
            *pp = se;
            return GS_IMG_OK;
        }
        default: break;                                    /* APPn, COM, ...: skipped */
        }
        p += len;
    }
}

int gs_jpeg_info(const uint8_t *data, size_t size, int *width, int *height, int *components) {
    if (!data) return GS_IMG_ERR_INVALID_ARGUMENT;
    Jpeg j;
    const uint8_t *scan;
    int rc = jpeg_begin(&j, data, size, &scan);
    if (rc == GS_IMG_OK) rc = next_scan(&j, &scan, data + size);
    if (rc < 0) return rc;
    if (rc == 1) return GS_IMG_ERR_CORRUPT;                /* no scan at all */
    if (width) *width = j.W;
    if (height) *height = j.H;
    if (components) *components = j.ncomp;
    return GS_IMG_OK;
}

/* ---- chroma upsampling ("fancy": triangle filter) ------------------------------------------------ */
/* one row, 2:1 horizontally; n input samples -> 2n outputs */
static void h2_fancy_row(const uint8_t *in, int n, uint8_t *out) {
    if (n == 1) {
        out[0] = out[1] = in[0];
        return;
    }
    int v = in[0];
    out[0] = (uint8_t)v;
    out[1] = (uint8_t)((v * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < n - 1; i++) {
        v = in[i] * 3;
        out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
        out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
    }
    v = in[n - 1];
    out[2 * n - 2] = (uint8_t)((v * 3 + in[n - 2] + 1) >> 2);
    out[2 * n - 1] = (uint8_t)v;
}
/* one output row of the 2:1 x 2:1 case from the nearer (in0) and the farther (in1) input row */
static void h2v2_fancy_row(const uint8_t *in0, const uint8_t *in1, int n, uint8_t *out) {
    int thiscol = in0[0] * 3 + in1[0];
    if (n == 1) {
        out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
        out[1] = (uint8_t)((thiscol * 4 + 7) >> 4);
        return;
    }
    int nextcol = in0[1] * 3 + in1[1], lastcol;
    out[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
    out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol;
    thiscol = nextcol;
    for (int i = 1; i < n - 1; i++) {
        nextcol = in0[i + 1] * 3 + in1[i + 1];
        out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
    }
    out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
}

/* Row `y` (0 <= y < H) of component c at full resolution -> dst[0 .. >= W). */
static void upsample_row(const Jpeg *j, const Comp *c, int y, uint8_t *dst, uint8_t *tmp) {
    const int hs = j->hmax / c->h, vs = j->vmax / c->v;
    const int stride = c->blocks_w * 8;
    const int n = c->ds_w;
    const int fancy = n > 2;     /* the library falls back to replication for very narrow planes */
    if (hs == 1 && vs == 1) {
        memcpy(dst, c->plane + (size_t)y * stride, (size_t)n);
        return;
    }
    if (hs == 2 && vs == 1 && fancy) {
        h2_fancy_row(c->plane + (size_t)y * stride, n, dst);
        return;
    }
    if (hs == 2 && vs == 2 && fancy) {
        const int r = y >> 1;
        int far_row = (y & 1) ? r + 1 : r - 1;      /* the row whose centre lies beyond this output row */
        if (far_row < 0) far_row = 0;               /* edges: the nearest real row stands in */
        if (far_row > c->ds_h - 1) far_row = c->ds_h - 1;
        h2v2_fancy_row(c->plane + (size_t)r * stride, c->plane + (size_t)far_row * stride, n, dst);
        return;
    }
    if (hs == 1 && vs == 2) {                        /* h1v2 "fancy": 3/4 - 1/4 vertically */
        const int r = y >> 1;
        int far_row = (y & 1) ? r + 1 : r - 1;
        if (far_row < 0) far_row = 0;
        if (far_row > c->ds_h - 1) far_row = c->ds_h - 1;
        const uint8_t *a = c->plane + (size_t)r * stride, *b = c->plane + (size_t)far_row * stride;
        const int bias = (y & 1) ? 2 : 1;
        for (int i = 0; i < n; i++) dst[i] = (uint8_t)((a[i] * 3 + b[i] + bias) >> 2);
        return;
    }
    /* any other integral ratio, and planes of one or two columns: sample replication */
    (void)tmp;
    const uint8_t *src = c->plane + (size_t)(y / vs) * stride;
    for (int i = 0; i < n; i++)
        for (int k = 0; k < hs; k++) dst[i * hs + k] = src[i];
}

/* ---- the decoder ----------------------------------------------------------------------------------- */
static int decode_block(Bits *b, const Huff *dc, const Huff *ac, int *pred, int16_t *coef) {
    memset(coef, 0, 64 * sizeof(int16_t));
    int t = decode_symbol(b, dc);
    if (t < 0 || t > 11) return GS_IMG_ERR_CORRUPT;
    int diff = t ? extend(get_bits(b, t), t) : 0;
    *pred += diff;
    coef[0] = (int16_t)*pred;
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(b, ac);
        if (rs < 0) return GS_IMG_ERR_CORRUPT;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r != 15) break;       /* EOB */
            k += 16;                  /* ZRL */
            continue;
        }
        k += r;
        if (k > 63) return GS_IMG_ERR_CORRUPT;
        coef[kZigzag[k]] = (int16_t)extend(get_bits(b, s), s);
        k++;
    }
    return GS_IMG_OK;
}

/* restart interval boundary: byte-align, expect RSTn */
static int take_restart(Bits *b, int *next_rst) {
    b->bitbuf = 0;
    b->bitcnt = 0;
    if (b->hit_marker) {
        if (b->hit_marker != 0xD0 + *next_rst) return GS_IMG_ERR_CORRUPT;
        b->p += 2;
        b->hit_marker = 0;
    } else {
        while (b->p + 1 < b->end && !(b->p[0] == 0xFF && b->p[1] >= 0xD0 && b->p[1] <= 0xD7)) b->p++;
        if (b->p + 1 >= b->end || b->p[1] != 0xD0 + *next_rst) return GS_IMG_ERR_CORRUPT;
        b->p += 2;
    }
    *next_rst = (*next_rst + 1) & 7;
    return GS_IMG_OK;
}

/* ---- progressive scans (T.81 annex G; the four procedures of the IJG library's jdphuff.c) ---------- */
typedef struct {
    int eobrun;                /* blocks still covered by the current end-of-band run */
} ProgState;

static int prog_dc_first(Bits *b, const Huff *dc, int *pred, int16_t *coef, int al) {
    const int t = decode_symbol(b, dc);
    if (t < 0 || t > 11) return GS_IMG_ERR_CORRUPT;
    *pred += t ? extend(get_bits(b, t), t) : 0;
    coef[0] = (int16_t)(*pred * (1 << al));
    return GS_IMG_OK;
}
static int prog_dc_refine(Bits *b, int16_t *coef, int al) {
    if (get_bits(b, 1)) coef[0] = (int16_t)(coef[0] | (1 << al));
    return GS_IMG_OK;
}
static int prog_ac_first(Bits *b, const Huff *ac, ProgState *st, int16_t *coef, int ss, int se, int al) {
    if (st->eobrun > 0) {
        st->eobrun--;
        return GS_IMG_OK;
    }
    for (int k = ss; k <= se; k++) {
        const int rs = decode_symbol(b, ac);
        if (rs < 0) return GS_IMG_ERR_CORRUPT;
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return GS_IMG_ERR_CORRUPT;
            coef[kZigzag[k]] = (int16_t)(extend(get_bits(b, s), s) * (1 << al));
        } else if (r == 15) {
            k += 15;                                       /* ZRL */
        } else {
            st->eobrun = 1 << r;                           /* EOBr: this block and eobrun - 1 more */
            if (r) st->eobrun += get_bits(b, r);
            st->eobrun--;
            break;
        }
    }
    return GS_IMG_OK;
}
static int prog_ac_refine(Bits *b, const Huff *ac, ProgState *st, int16_t *coef, int ss, int se, int al) {
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (st->eobrun == 0) {
        for (; k <= se; k++) {
            const int rs = decode_symbol(b, ac);
            if (rs < 0) return GS_IMG_ERR_CORRUPT;
            int r = rs >> 4, s = rs & 15;
            if (s) {
                if (s != 1) return GS_IMG_ERR_CORRUPT;     /* a newly non-zero coefficient is +-1 */
                s = get_bits(b, 1) ? p1 : m1;
            } else if (r != 15) {
                st->eobrun = 1 << r;                       /* EOBr: the band ends here */
                if (r) st->eobrun += get_bits(b, r);
                break;
            }
            /* pass over r still-zero coefficients, correcting every already non-zero one on the way */
            do {
                int16_t *c = &coef[kZigzag[k]];
                if (*c != 0) {
                    if (get_bits(b, 1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                } else if (--r < 0) {
                    break;
                }
                k++;
            } while (k <= se);
            if (s) {
                if (k > 63) return GS_IMG_ERR_CORRUPT;
                coef[kZigzag[k]] = (int16_t)s;
            }
        }
    }
    if (st->eobrun > 0) {
        /* inside an end-of-band run: only correction bits for the non-zero coefficients that remain */
        for (; k <= se; k++) {
            int16_t *c = &coef[kZigzag[k]];
            if (*c != 0 && get_bits(b, 1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
        }
        st->eobrun--;
    }
    return GS_IMG_OK;
}

/* One scan of a multi-scan file into the coefficient arrays (coefs[i]: blocks_w * blocks_h * 64, natural
 * order).  Leaves *pp at the marker that ends the entropy-coded segment. */
static int decode_scan_to_coefs(Jpeg *j, int16_t **coefs, const uint8_t **pp, const uint8_t *end) {
    const int prog = j->progressive;
    const int ss = prog ? j->ss : 0, se = prog ? j->se : 63, ah = prog ? j->ah : 0, al = prog ? j->al : 0;
    if (prog) {
        /* G.1.1.1.1: DC scans carry no AC band; an AC scan has one component */
        if (ss > se || se > 63 || al > 13 || ah > 13) return GS_IMG_ERR_CORRUPT;
        if (ss == 0 && se != 0) return GS_IMG_ERR_CORRUPT;
        if (ss > 0 && j->ns != 1) return GS_IMG_ERR_CORRUPT;
    }
    for (int i = 0; i < j->ns; i++) {
        const Comp *c = &j->c[j->scomp[i]];
        const int need_dc = ss == 0 && ah == 0, need_ac = se > 0;
        if ((need_dc && !j->dc[c->td].present) || (need_ac && !j->ac[c->ta].present)) return GS_IMG_ERR_CORRUPT;
    }
    Bits b;
    memset(&b, 0, sizeof(b));
    b.p = *pp;
    b.end = end;
    ProgState st = {0};
    int restart_left = j->restart_interval, next_rst = 0, rc = GS_IMG_OK;
    for (int i = 0; i < j->ncomp; i++) j->c[i].pred = 0;
    const int mcu_w = 8 * j->hmax, mcu_h = 8 * j->vmax;
    const int interleaved = j->ns > 1;
    const Comp *c0 = &j->c[j->scomp[0]];
    /* a one-component scan walks that component's own blocks (T.81 A.2.2), an interleaved one the MCUs */
    const int total_x = interleaved ? (j->W + mcu_w - 1) / mcu_w : (c0->ds_w + 7) / 8;
    const int total_y = interleaved ? (j->H + mcu_h - 1) / mcu_h : (c0->ds_h + 7) / 8;
    for (int my = 0; my < total_y && rc == GS_IMG_OK; my++)
        for (int mx = 0; mx < total_x && rc == GS_IMG_OK; mx++) {
            if (j->restart_interval && restart_left == 0) {
                rc = take_restart(&b, &next_rst);
                if (rc) break;
                restart_left = j->restart_interval;
                for (int i = 0; i < j->ncomp; i++) j->c[i].pred = 0;
                st.eobrun = 0;
            }
            for (int i = 0; i < j->ns && rc == GS_IMG_OK; i++) {
                Comp *c = &j->c[j->scomp[i]];
                const int h = interleaved ? c->h : 1, v = interleaved ? c->v : 1;
                for (int by = 0; by < v && rc == GS_IMG_OK; by++)
                    for (int bx = 0; bx < h && rc == GS_IMG_OK; bx++) {
                        const size_t blk = (size_t)(my * v + by) * c->blocks_w + (size_t)(mx * h + bx);
                        int16_t *coef = coefs[j->scomp[i]] + blk * 64;
                        if (!prog) rc = decode_block(&b, &j->dc[c->td], &j->ac[c->ta], &c->pred, coef);
                        else if (ss == 0) rc = ah == 0 ? prog_dc_first(&b, &j->dc[c->td], &c->pred, coef, al)
                                                       : prog_dc_refine(&b, coef, al);
                        else rc = ah == 0 ? prog_ac_first(&b, &j->ac[c->ta], &st, coef, ss, se, al)
                                          : prog_ac_refine(&b, &j->ac[c->ta], &st, coef, ss, se, al);
                    }
            }
            restart_left--;
        }
    if (rc) return rc;
    /* the marker behind the entropy-coded data (the bit reader never reads past one) */
    const uint8_t *p = b.p;
    if (!b.hit_marker)
        while (p + 1 < end && !(p[0] == 0xFF && p[1] != 0x00 && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
    *pp = p;
    return GS_IMG_OK;
}

/* sample planes -> interleaved RGB: fancy upsampling + fixed-point YCbCr -> RGB */
static int planes_to_rgb(const Jpeg *j, uint8_t *out) {
    /* colour conversion tables (16-bit fixed point), per call and on the stack: 4 KiB and 256
     * iterations are nothing next to a decode, and a decoder called from several loader threads
     * (ctypes releases the GIL) must not share lazily initialised statics */
    int crr[256], cbb[256];
    long crg[256], cbg[256];
    for (int i = 0; i < 256; i++) {
        const long x = i - 128;
        crr[i] = (int)((91881L * x + 32768L) >> 16);      /* FIX(1.40200) */
        cbb[i] = (int)((116130L * x + 32768L) >> 16);     /* FIX(1.77200) */
        crg[i] = -46802L * x;                             /* FIX(0.71414) */
        cbg[i] = -22554L * x + 32768L;                    /* FIX(0.34414), + ONE_HALF */
    }
    const int mcus_x = (j->W + 8 * j->hmax - 1) / (8 * j->hmax);
    const size_t roww = (size_t)(mcus_x * 8 * j->hmax + 16);
    uint8_t *rows = (uint8_t *)malloc(roww * 4);
    if (!rows) return GS_IMG_ERR_INVALID_ARGUMENT;
    const int ycc = j->ncomp == 3 && j->adobe_transform != 0;   /* Adobe transform 0: stored as RGB */
    for (int y = 0; y < j->H; y++) {
        uint8_t *o = out + (size_t)y * j->W * 3;
        if (j->ncomp == 1) {
            const uint8_t *g = j->c[0].plane + (size_t)y * j->c[0].blocks_w * 8;
            for (int x = 0; x < j->W; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = g[x];
            continue;
        }
        uint8_t *r0 = rows, *r1 = rows + roww, *r2 = rows + 2 * roww;
        upsample_row(j, &j->c[0], y, r0, rows + 3 * roww);
        upsample_row(j, &j->c[1], y, r1, rows + 3 * roww);
        upsample_row(j, &j->c[2], y, r2, rows + 3 * roww);
        if (!ycc) {
            for (int x = 0; x < j->W; x++) { o[3 * x] = r0[x]; o[3 * x + 1] = r1[x]; o[3 * x + 2] = r2[x]; }
            continue;
        }
        for (int x = 0; x < j->W; x++) {
            const int Y = r0[x], cb = r1[x], cr = r2[x];
            int R = Y + crr[cr], G = Y + (int)((cbg[cb] + crg[cr]) >> 16), B = Y + cbb[cb];
            o[3 * x] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R));
            o[3 * x + 1] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G));
            o[3 * x + 2] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
        }
    }
    free(rows);
    return GS_IMG_OK;
}

int gs_jpeg_decode_rgb(const uint8_t *data, size_t size, uint8_t *out, size_t out_bytes) {
    if (!data || !out) return GS_IMG_ERR_INVALID_ARGUMENT;
    Jpeg j;
    const uint8_t *scan, *end = data + size;
    int rc = jpeg_begin(&j, data, size, &scan);
    if (rc == GS_IMG_OK) rc = next_scan(&j, &scan, end);
    if (rc < 0) return rc;
    if (rc == 1) return GS_IMG_ERR_CORRUPT;                /* no scan */
    if (out_bytes < (size_t)j.W * j.H * 3) return GS_IMG_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < j.ncomp; i++) {
        if (j.hmax % j.c[i].h || j.vmax % j.c[i].v) return GS_IMG_ERR_UNSUPPORTED;  /* fractional ratios */
        if (!j.qt_present[j.c[i].tq] && !j.progressive) return GS_IMG_ERR_CORRUPT;
    }
    const int mcu_w = 8 * j.hmax, mcu_h = 8 * j.vmax;
    const int mcus_x = (j.W + mcu_w - 1) / mcu_w, mcus_y = (j.H + mcu_h - 1) / mcu_h;
    for (int i = 0; i < j.ncomp; i++) {
        Comp *c = &j.c[i];
        /* a single-component scan is non-interleaved: its MCU is one block (T.81 A.2.2) */
        const int h = j.ncomp == 1 ? 1 : c->h, v = j.ncomp == 1 ? 1 : c->v;
        c->ds_w = (j.W * c->h + j.hmax - 1) / j.hmax;
        c->ds_h = (j.H * c->v + j.vmax - 1) / j.vmax;
        if (j.ncomp == 1) {
            c->blocks_w = (c->ds_w + 7) / 8;
            c->blocks_h = (c->ds_h + 7) / 8;
        } else {
            c->blocks_w = mcus_x * h;
            c->blocks_h = mcus_y * v;
        }
        c->plane = (uint8_t *)malloc((size_t)c->blocks_w * 8 * c->blocks_h * 8);
        if (!c->plane) {
            for (int k = 0; k < i; k++) free(j.c[k].plane);
            return GS_IMG_ERR_INVALID_ARGUMENT;
        }
        c->pred = 0;
    }
    rc = GS_IMG_OK;
    if (!j.progressive && j.ns == j.ncomp) {
        /* ---- one sequential scan with every component: MCU by MCU straight into the planes ---- */
        for (int i = 0; i < j.ncomp; i++) {
            const Comp *c = &j.c[i];
            if (!j.dc[c->td].present || !j.ac[c->ta].present) rc = GS_IMG_ERR_CORRUPT;
        }
        Bits b;
        memset(&b, 0, sizeof(b));
        b.p = scan;
        b.end = end;
        int16_t coef[64];
        const int single = j.ncomp == 1;
        const int total_x = single ? j.c[0].blocks_w : mcus_x, total_y = single ? j.c[0].blocks_h : mcus_y;
        int restart_left = j.restart_interval, next_rst = 0;
        for (int my = 0; my < total_y && rc == GS_IMG_OK; my++) {
            for (int mx = 0; mx < total_x && rc == GS_IMG_OK; mx++) {
                if (j.restart_interval && restart_left == 0) {
                    rc = take_restart(&b, &next_rst);
                    if (rc) break;
                    restart_left = j.restart_interval;
                    for (int i = 0; i < j.ncomp; i++) j.c[i].pred = 0;
                }
                for (int i = 0; i < j.ncomp && rc == GS_IMG_OK; i++) {
                    Comp *c = &j.c[i];
                    const int h = single ? 1 : c->h, v = single ? 1 : c->v;
                    const int stride = c->blocks_w * 8;
                    for (int by = 0; by < v && rc == GS_IMG_OK; by++)
                        for (int bx = 0; bx < h; bx++) {
                            rc = decode_block(&b, &j.dc[c->td], &j.ac[c->ta], &c->pred, coef);
                            if (rc) break;
                            uint8_t *o = c->plane + ((size_t)(my * v + by) * 8) * stride + (size_t)(mx * h + bx) * 8;
                            idct_islow(coef, j.qt[c->tq], o, stride);
                        }
                }
                restart_left--;
            }
        }
    } else {
        /* ---- several scans (progressive, or a sequential file with one scan per component): all
         *      coefficients first, the transform at the end ---- */
        int16_t *coefs[MAX_COMPS] = {NULL, NULL, NULL};
        for (int i = 0; i < j.ncomp && rc == GS_IMG_OK; i++) {
            coefs[i] = (int16_t *)calloc((size_t)j.c[i].blocks_w * j.c[i].blocks_h * 64, sizeof(int16_t));
            if (!coefs[i]) rc = GS_IMG_ERR_INVALID_ARGUMENT;
        }
        int more = 1, scans = 0;
        while (rc == GS_IMG_OK && more) {
            /* a progressive file has about ten scans; a crafted one with thousands of tiny scans would make
             * the decoder walk the whole coefficient array once per scan: capped */
            if (++scans > 1000) { rc = GS_IMG_ERR_CORRUPT; break; }
            rc = decode_scan_to_coefs(&j, coefs, &scan, end);
            if (rc) break;
            const int nx = next_scan(&j, &scan, end);
            if (nx < 0) rc = nx;
            else more = nx == GS_IMG_OK;
        }
        for (int i = 0; i < j.ncomp && rc == GS_IMG_OK; i++) {
            const Comp *c = &j.c[i];
            if (!j.qt_present[c->tq]) { rc = GS_IMG_ERR_CORRUPT; break; }   /* (tables may follow the frame header) */
            const int stride = c->blocks_w * 8;
            for (int by = 0; by < c->blocks_h; by++)
                for (int bx = 0; bx < c->blocks_w; bx++)
                    idct_islow(coefs[i] + ((size_t)by * c->blocks_w + bx) * 64, j.qt[c->tq],
                               c->plane + ((size_t)by * 8) * stride + (size_t)bx * 8, stride);
        }
        for (int i = 0; i < MAX_COMPS; i++) free(coefs[i]);
    }
    if (rc == GS_IMG_OK) rc = planes_to_rgb(&j, out);
    for (int i = 0; i < j.ncomp; i++) free(j.c[i].plane);
    return rc;
}

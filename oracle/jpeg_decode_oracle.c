/*
 * jpeg_decode_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * CPU restatement of a baseline JPEG decoder, the checker of ultragrid_amd/csrc/jpeg_decode.hip (receive side of the JPEG path:
 * src/video_decompress/gpujpeg.c:74-140 configures, :292-301 calls gpujpeg_decoder_decode of the external libgpujpeg).
 *
 * What it restates is published: ITU-T T.81 (marker syntax B.2, Huffman decoding F.2.2 with the table generation of Annex C,
 * restart intervals E.2.4 / F.2.1.3, byte stuffing B.1.1.5) and the inverse DCT every libjpeg since release 6 uses by default
 * (jidctint.c "slow but accurate integer", Loeffler-Ligtenberg-Moschytz with 13-bit constants, PASS1_BITS 2).
 *
 * PARITY: the reference's decoder lives in external libgpujpeg (not under /root/reference): unpinned against it.  PINNED instead to
 * libjpeg-turbo as shipped with Pillow (tests/test_oracle_jpeg_decode.py): the component planes this file produces equal libjpeg's
 * bit for bit wherever libjpeg hands out untouched samples -- all three planes of R,G,B (Adobe transform 0) and 4:4:4 YCbCr streams
 * (Pillow's "YCbCr" draft mode), the luma plane of 4:2:2 / 4:2:0 streams -- with and without restart intervals, on streams written by
 * libjpeg and by this repository's encoder.
 *
 * Output = the component planes at their own resolution (8-bit, padded to whole MCUs), nothing else: upsampling / colour conversion /
 * packing are the product's existing pixel-format kernels, checked elsewhere.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct {
        uint8_t bits[17];       /* number of codes of each length 1..16 */
        uint8_t vals[256];
        int mincode[17], maxcode[18], valptr[17];
        int present;
} huff_t;

typedef struct {
        const uint8_t *p, *end;
        uint32_t acc;
        int nbits;
        int hit_marker; /* stopped at a marker: further bits are zeros */
} bitreader_t;

static void build_huff(huff_t *h)
{
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
                h->valptr[l] = k;
                h->mincode[l] = code;
                code += h->bits[l];
                k += h->bits[l];
                h->maxcode[l] = h->bits[l] ? code - 1 : -1;
                code <<= 1;
        }
        h->maxcode[17] = 0x7fffffff;
}

static int next_bit(bitreader_t *b)
{
        if (b->nbits == 0) {
                uint32_t byte = 0;
                if (!b->hit_marker && b->p < b->end) {
                        byte = *b->p;
                        if (byte == 0xFF) {
                                if (b->p + 1 < b->end && b->p[1] == 0x00) {
                                        b->p += 2; /* stuffed zero */
                                } else {
                                        b->hit_marker = 1; /* RSTn / EOI: the segment is over, feed zeros */
                                        byte = 0;
                                }
                        } else {
                                b->p++;
                        }
                }
                b->acc = byte;
                b->nbits = 8;
        }
        b->nbits--;
        return (int) ((b->acc >> b->nbits) & 1u);
}

static int receive(bitreader_t *b, int n)
{
        int v = 0;
        for (int i = 0; i < n; i++) v = (v << 1) | next_bit(b);
        return v;
}

static int decode_symbol(bitreader_t *b, const huff_t *h)
{
        int code = next_bit(b), l = 1;
        while (l <= 16 && (h->maxcode[l] < 0 || code > h->maxcode[l])) {
                code = (code << 1) | next_bit(b);
                l++;
        }
        if (l > 16) return 0; /* corrupt data */
        return h->vals[h->valptr[l] + code - h->mincode[l]];
}

static int extend(int v, int t) { return t && v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

static const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                     35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

/* jidctint.c (libjpeg 6b and later; libjpeg-turbo's SIMD forms are bit-exact with it): dequantised coefficients in natural order
 * -> 64 samples, +128, clamped */
#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
static void idct_islow(const int *in, uint8_t *out, int out_pitch)
{
        int ws[64];
        for (int pass = 0; pass < 2; pass++) {
                for (int i = 0; i < 8; i++) {
                        const int *s = pass == 0 ? in + i : ws + 8 * i;
                        const int st = pass == 0 ? 8 : 1;
                        int z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
                        z2 = s[2 * st]; z3 = s[6 * st];
                        z1 = (z2 + z3) * 4433;
                        tmp2 = z1 + z3 * (-15137);
                        tmp3 = z1 + z2 * 6270;
                        z2 = s[0]; z3 = s[4 * st];
                        tmp0 = (z2 + z3) * (1 << CONST_BITS);
                        tmp1 = (z2 - z3) * (1 << CONST_BITS);
                        tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
                        tmp0 = s[7 * st]; tmp1 = s[5 * st]; tmp2 = s[3 * st]; tmp3 = s[1 * st];
                        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
                        z5 = (z3 + z4) * 9633;
                        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
                        z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
                        z3 += z5; z4 += z5;
                        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
                        const int o[8] = { tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3 };
                        for (int k = 0; k < 8; k++) {
                                if (pass == 0) {
                                        ws[8 * k + i] = DESCALE(o[k], CONST_BITS - PASS1_BITS);
                                } else {
                                        int v = DESCALE(o[k], CONST_BITS + PASS1_BITS + 3) + 128;
                                        out[i * out_pitch + k] = (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v));
                                }
                        }
                }
        }
}

/* Decodes a baseline stream.  info = { width, height, components, h0, v0, h1, v1, h2, v2, restart_interval, adobe_transform (-1: no
 * Adobe marker), scans }.  planes[c] receives component c, pitch[c] bytes per line, (MCU-padded) -- the caller allocates
 * ceil-to-MCU sizes: plane c is (mcu_w * 8 * h_c) x (mcu_h * 8 * v_c).  planes may be NULL to query `info` only.
 * Returns 0, or a negative code: -1 not a baseline JPEG this decoder takes, -2 truncated. */
static int decode_impl(const uint8_t *data, long len, int info[12], uint8_t *planes[3], const int pitch[3], int16_t *qcoef[3])
{
        uint16_t qt[4][64];
        huff_t dc[4], ac[4];
        memset(dc, 0, sizeof dc);
        memset(ac, 0, sizeof ac);
        int width = 0, height = 0, ncomp = 0, ri = 0, adobe = -1, scans = 0;
        int hs[3] = { 1, 1, 1 }, vs[3] = { 1, 1, 1 }, tq[3] = { 0, 0, 0 }, cid[3] = { 0, 0, 0 };
        long pos = 2;
        if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return -1;
        int *coef = NULL; /* per component: blocks in raster order of the component's (MCU-padded) block grid, 64 ints each, natural order, dequantised */
        long coef_off[3] = { 0, 0, 0 };
        int mcu_w = 0, mcu_h = 0, hmax = 1, vmax = 1;
        int rc = 0;
        while (pos + 4 <= len) {
                if (data[pos] != 0xFF) { rc = -1; break; }
                const int m = data[pos + 1];
                if (m == 0xD9) break;
                const long seglen = (data[pos + 2] << 8) | data[pos + 3];
                const uint8_t *s = data + pos + 4;
                if (pos + 2 + seglen > len) { rc = -2; break; }
                if (m == 0xDB) {
                        for (long o = 0; o + 65 <= seglen - 2;) {
                                const int pq = s[o] >> 4, t = s[o] & 15;
                                if (pq != 0 || t > 3) { rc = -1; break; }
                                for (int k = 0; k < 64; k++) qt[t][kZigzag[k]] = s[o + 1 + k];
                                o += 65;
                        }
                } else if (m == 0xC0) {
                        if (s[0] != 8) { rc = -1; break; }
                        height = (s[1] << 8) | s[2]; width = (s[3] << 8) | s[4]; ncomp = s[5];
                        if (ncomp != 1 && ncomp != 3) { rc = -1; break; }
                        for (int c = 0; c < ncomp; c++) {
                                cid[c] = s[6 + 3 * c]; hs[c] = s[7 + 3 * c] >> 4; vs[c] = s[7 + 3 * c] & 15; tq[c] = s[8 + 3 * c];
                                if (hs[c] > hmax) hmax = hs[c];
                                if (vs[c] > vmax) vmax = vs[c];
                        }
                } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
                        rc = -1; /* not baseline */
                        break;
                } else if (m == 0xC4) {
                        for (long o = 0; o + 17 <= seglen - 2;) {
                                const int tc = s[o] >> 4, th = s[o] & 15;
                                if (th > 3 || tc > 1) { rc = -1; break; }
                                huff_t *h = tc ? &ac[th] : &dc[th];
                                int n = 0;
                                h->bits[0] = 0;
                                for (int l = 1; l <= 16; l++) n += (h->bits[l] = s[o + l]);
                                if (n > 256) { rc = -1; break; }
                                memcpy(h->vals, s + o + 17, (size_t) n);
                                build_huff(h);
                                h->present = 1;
                                o += 17 + n;
                        }
                } else if (m == 0xDD) {
                        ri = (s[0] << 8) | s[1];
                } else if (m == 0xEE && seglen >= 14 && memcmp(s, "Adobe", 5) == 0) {
                        adobe = s[11];
                } else if (m == 0xDA) {
                        if (!width || !ncomp) { rc = -1; break; }
                        mcu_w = (width + 8 * hmax - 1) / (8 * hmax); mcu_h = (height + 8 * vmax - 1) / (8 * vmax);
                        if (!coef && planes) {
                                long total = 0;
                                for (int c = 0; c < ncomp; c++) { coef_off[c] = total; total += (long) mcu_w * hs[c] * mcu_h * vs[c] * 64; }
                                coef = calloc((size_t) total, sizeof *coef);
                        }
                        const int ns = s[0];
                        int sc[3], td[3], ta[3];
                        for (int k = 0; k < ns; k++) {
                                sc[k] = -1;
                                for (int c = 0; c < ncomp; c++) if (cid[c] == s[1 + 2 * k]) sc[k] = c;
                                td[k] = s[2 + 2 * k] >> 4; ta[k] = s[2 + 2 * k] & 15;
                                if (sc[k] < 0) rc = -1;
                        }
                        if (rc) break;
                        scans++;
                        /* entropy-coded data follows the SOS header */
                        const uint8_t *p = data + pos + 2 + seglen, *const end = data + len;
                        /* a non-interleaved scan (one component) walks that component's own block grid, ceil(size / 8) blocks (A.2.2) */
                        const int single = ns == 1;
                        const int c0 = sc[0];
                        const int bw1 = single ? ((width * hs[c0] + hmax - 1) / hmax + 7) / 8 : 0, bh1 = single ? ((height * vs[c0] + vmax - 1) / vmax + 7) / 8 : 0;
                        const long units = single ? (long) bw1 * bh1 : (long) mcu_w * mcu_h;
                        int pred[3] = { 0, 0, 0 };
                        bitreader_t br = { p, end, 0, 0, 0 };
                        for (long u = 0; u < units; u++) {
                                if (ri && u && u % ri == 0) { /* restart: skip to just behind the next RSTn marker */
                                        const uint8_t *q = br.p;
                                        while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) q++;
                                        br.p = q + 2 <= end ? q + 2 : end;
                                        br.nbits = 0; br.hit_marker = 0;
                                        pred[0] = pred[1] = pred[2] = 0;
                                }
                                for (int k = 0; k < ns; k++) {
                                        const int c = sc[k];
                                        const int nb_h = single ? 1 : hs[c], nb_v = single ? 1 : vs[c];
                                        for (int by = 0; by < nb_v; by++) {
                                                for (int bx = 0; bx < nb_h; bx++) {
                                                        int blk[64] = { 0 };
                                                        int16_t coded[64] = { 0 }; /* the quantised values as the stream holds them, zig-zag order */
                                                        const int t = decode_symbol(&br, &dc[td[k]]);
                                                        pred[k] += extend(receive(&br, t), t);
                                                        coded[0] = (int16_t) pred[k];
                                                        /* a coefficient is 16 bits wide (libjpeg's JCOEF; jdhuff.c stores `(JCOEF) s`): only damaged
                                                         * streams ever run the DC prediction out of that range */
                                                        blk[0] = (int16_t) pred[k] * qt[tq[c]][0];
                                                        for (int z = 1; z < 64;) {
                                                                const int rs = decode_symbol(&br, &ac[ta[k]]);
                                                                const int r = rs >> 4, sz = rs & 15;
                                                                if (sz == 0) {
                                                                        if (r != 15) break; /* EOB */
                                                                        z += 16;
                                                                        continue;
                                                                }
                                                                z += r;
                                                                /* the extra bits are read before the position is looked at, as in libjpeg (jdhuff.c: k += r;
                                                                 * r = GET_BITS(s) ...): a run that leaves the block (damaged streams only) still eats them */
                                                                const int value = extend(receive(&br, sz), sz);
                                                                if (z > 63) break;
                                                                blk[kZigzag[z]] = value * qt[tq[c]][kZigzag[z]];
                                                                coded[z] = (int16_t) value;
                                                                z++;
                                                        }
                                                        if (coef || qcoef) {
                                                                const long gw = (long) mcu_w * hs[c];
                                                                long bxg, byg;
                                                                if (single) { bxg = u % bw1; byg = u / bw1; }
                                                                else { bxg = (u % mcu_w) * hs[c] + bx; byg = (u / mcu_w) * vs[c] + by; }
                                                                if (coef) memcpy(coef + coef_off[c] + (byg * gw + bxg) * 64, blk, sizeof blk);
                                                                if (qcoef && qcoef[c]) memcpy(qcoef[c] + (byg * gw + bxg) * 64, coded, sizeof coded);
                                                        }
                                                }
                                        }
                                }
                        }
                        /* continue behind the scan: find the next marker that is not RSTn / stuffing */
                        const uint8_t *q = br.p;
                        while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
                        pos = q - data;
                        continue;
                }
                if (rc) break;
                pos += 2 + seglen;
        }
        if (rc == 0 && (!width || !scans)) rc = -1;
        if (info) {
                const int v[12] = { width, height, ncomp, hs[0], vs[0], hs[1], vs[1], hs[2], vs[2], ri, adobe, scans };
                memcpy(info, v, sizeof v);
        }
        if (rc == 0 && planes && coef) {
                for (int c = 0; c < ncomp; c++) {
                        const long gw = (long) mcu_w * hs[c], gh = (long) mcu_h * vs[c];
                        for (long by = 0; by < gh; by++) {
                                for (long bx = 0; bx < gw; bx++) idct_islow(coef + coef_off[c] + (by * gw + bx) * 64, planes[c] + by * 8 * pitch[c] + bx * 8, pitch[c]);
                        }
                }
        }
        free(coef);
        return rc;
}

int oracle_jpeg_decode(const uint8_t *data, long len, int info[12], uint8_t *planes[3], const int pitch[3])
{
        return decode_impl(data, len, info, planes, pitch, NULL);
}

/* The entropy decoder alone: qcoef[c] receives the QUANTISED coefficients of component c exactly as the stream codes them -- blocks in
 * raster order of the component's MCU-padded block grid ((mcu_w * h_c) x (mcu_h * v_c) blocks), 64 int16 each in zig-zag order: the layout
 * of oracle_jpeg_fdct_quant_plane's output, so that "what the encoder wrote" can be compared with "what the FDCT oracle computes"
 * coefficient by coefficient (bench.py's parity_check of the JPEG encoder leg). */
int oracle_jpeg_decode_coeffs(const uint8_t *data, long len, int info[12], int16_t *qcoef[3])
{
        return decode_impl(data, len, info, NULL, NULL, qcoef);
}

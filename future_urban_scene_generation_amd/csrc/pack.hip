// Host-side weight pre-packing through the C ABI (fusg_pack_conv_sizes / fusg_pack_conv_weights): a caller that is not
// Python can drive fusg_conv2d end to end.  Produces, for one nn.Conv2d-style filter [cout][cin][kh][kw] (optionally read
// from two concatenated sources), exactly what future_urban_scene_generation_amd/pack.py produces: the K-contiguous
// panel, the k-table of the table-driven gather, the padded bias, the scaled split-fp16 (hi, lo) panels with their
// per-channel inverse scale, and - when the layer qualifies - the MFMA-fragment-order copy for the halo kernel (plain
// or parity-quadrant order) or the per-step order of the tap-unit kernel.  tests/test_pack_cpu.py checks the two
// implementations against each other bit for bit.  No device work: pure host code.
#include <math.h>
#include <vector>
#include "common.h"

namespace fusg {

static inline int ru(int x, int m) { return (x + m - 1) / m * m; }

struct PackGeom {
    int taps, c0, c1, c0k, c1k, ctot, k, k_pad, cout_pad;
    bool frag_ok, s2d_ok, tap_ok;
    int unit, nsteps;
};

static bool geom(const fusg_pack_spec& s, PackGeom& g) {
    if (s.cout <= 0 || s.cin <= 0 || s.kh <= 0 || s.kw <= 0 || s.c0 <= 0 || s.c0 > s.cin || s.dil < 1 || s.pad < 0 ||
        (s.stride != 1 && s.stride != 2) || (s.upsample != 0 && s.upsample != 1) || s.cin_pad < 4 || s.cin_pad % 4) return false;
    g.taps = s.kh * s.kw;
    g.c0 = s.c0; g.c1 = s.cin - s.c0;
    g.c0k = ru(g.c0, s.cin_pad); g.c1k = g.c1 ? ru(g.c1, s.cin_pad) : 0;
    g.ctot = g.c0k + g.c1k;
    g.k = g.taps * g.ctot;
    g.k_pad = ru(g.k, 32);
    g.cout_pad = ru(s.cout, 32);
    g.frag_ok = g.c0k % 32 == 0 && g.c1k % 32 == 0 && g.ctot > 0 && g.k_pad == g.taps * g.ctot;
    g.s2d_ok = s.stride == 2 && s.kh == s.kw && (s.kh == 3 || s.kh == 4) && s.pad == 1 && s.dil == 1 && s.upsample == 0 &&
               g.c1k == 0 && g.c0k % 32 == 0 && g.c0k > 0;
    g.unit = g.c0k % 8 == 0 ? 8 : 4;
    g.tap_ok = (g.taps >= 9 || g.c0k <= 8) && g.c1k == 0 && g.c0k >= 4 && g.c0k <= 24 && s.dil == 1 && s.upsample == 0 &&
               g.k_pad >= g.taps * g.c0k && g.taps * (g.c0k / g.unit) <= 160;
    g.nsteps = (g.taps * g.c0k + 15) / 16;
    return true;
}

static inline uint16_t f16_bits(float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; }
static inline double f16_val(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (double)(float)h; }

}  // namespace fusg

using namespace fusg;

extern "C" int fusg_pack_conv_sizes(const fusg_pack_spec* s, fusg_pack_sizes* o) {
    PackGeom g;
    FUSG_CHECK(s && o && geom(*s, g), "pack_conv_sizes: bad spec");
    memset(o, 0, sizeof(*o));
    o->cout_pad = g.cout_pad; o->k_pad = g.k_pad; o->c0k = g.c0k; o->c1k = g.c1k;
    o->wpack_floats = (int64_t)g.cout_pad * g.k_pad;
    o->ktab_ints = (int64_t)(g.k_pad / 4) * 2;
    o->wpack_h_halves = 2 * o->wpack_floats;
    o->wfrag_order = -1;
    if (g.frag_ok) { o->wfrag_order = g.s2d_ok ? 1 : 0; o->wfrag_halves = o->wpack_h_halves; }
    else if (g.tap_ok) { o->wfrag_order = 2; o->wfrag_halves = (int64_t)g.nsteps * (g.cout_pad / 32) * 2 * 64 * 8; }
    return FUSG_OK;
}

extern "C" int fusg_pack_conv_weights(const fusg_pack_spec* s, const float* weight, const float* bias, float* wpack, int32_t* ktab,
                                      float* bias_pad, uint16_t* wpack_h, float* wscale, uint16_t* wfrag) {
    PackGeom g;
    FUSG_CHECK(s && weight && wpack && ktab && bias_pad && wpack_h && wscale && geom(*s, g), "pack_conv_weights: bad spec / null buffer");
    const int taps = g.taps, ctot = g.ctot, K = g.k_pad, CP = g.cout_pad;
    // ---- panel [cout_pad][k_pad], K order (tap, concat channel), and the k-table
    for (long i = 0; i < (long)CP * K; ++i) wpack[i] = 0.f;
    for (int co = 0; co < s->cout; ++co)
        for (int c = 0; c < s->cin; ++c) {
            const int kc = c < g.c0 ? c : g.c0k + (c - g.c0);
            for (int t = 0; t < taps; ++t)
                wpack[(long)co * K + t * ctot + kc] = weight[((long)co * s->cin + c) * taps + t];
        }
    for (int t = 0; t < taps; ++t) {
        const int ky = t / s->kw, kx = t - ky * s->kw, dy = ky * s->dil - s->pad, dx = kx * s->dil - s->pad;
        for (int q = 0; q < ctot / 4; ++q) {
            const int cc = q * 4, src = cc < g.c0k ? 0 : 1, coff = src ? cc - g.c0k : cc;
            int32_t* e = ktab + ((long)(t * ctot + cc) / 4) * 2;
            e[0] = (int32_t)(((uint32_t)dy & 0xFFFFu) | (((uint32_t)dx & 0xFFFFu) << 16));
            e[1] = (int32_t)(((uint32_t)coff & 0x3FFFFFFFu) | ((uint32_t)src << 30));
        }
    }
    for (int q = g.k / 4; q < K / 4; ++q) { ktab[q * 2] = 0; ktab[q * 2 + 1] = (int32_t)0x80000000u; }
    for (int n = 0; n < CP; ++n) bias_pad[n] = (bias && n < s->cout) ? bias[n] : 0.f;
    // ---- scaled split: per output channel the power of two that puts the largest |w| in [2^13, 2^14)
    for (int n = 0; n < CP; ++n) {
        double rowmax = 0.0;
        for (int k = 0; k < K; ++k) rowmax = fmax(rowmax, fabs((double)wpack[(long)n * K + k]));
        int sh = 0;
        if (isfinite(rowmax) && rowmax > 0.0) {
            int e;
            (void)frexp(rowmax, &e);
            sh = 14 - e;
            sh = sh < -100 ? -100 : (sh > 100 ? 100 : sh);
        }
        const double sc = ldexp(1.0, sh);
        wscale[n] = (float)(1.0 / sc);
        for (int k = 0; k < K; ++k) {
            const double ws = (double)wpack[(long)n * K + k] * sc;
            const uint16_t hi = f16_bits((float)ws);
            const uint16_t lo = f16_bits((float)(ws - f16_val(hi)));
            wpack_h[(long)n * K + k] = hi;
            wpack_h[(long)CP * K + (long)n * K + k] = lo;
        }
    }
    if (!wfrag) return FUSG_OK;
    const int nt32 = CP / 32;
    if (g.frag_ok) {
        // [tap][chunk32][nt32][16-column half][hi|lo][lane = g*16 + r][8]; parity-quadrant tap order for stride-2 k3/k4
        const int nch = ctot / 32;
        std::vector<int> order(taps);
        if (g.s2d_ok) {
            int n = 0;
            for (int q = 0; q < 4; ++q)
                for (int ky = 0; ky < s->kh; ++ky) {
                    if ((((ky - 1) % 2) + 2) % 2 != (q >> 1)) continue;
                    for (int kx = 0; kx < s->kw; ++kx)
                        if ((((kx - 1) % 2) + 2) % 2 == (q & 1)) order[n++] = ky * s->kw + kx;
                }
        } else {
            for (int t = 0; t < taps; ++t) order[t] = t;
        }
        long o = 0;
        for (int t = 0; t < taps; ++t)
            for (int ch = 0; ch < nch; ++ch)
                for (int nt = 0; nt < nt32; ++nt)
                    for (int ct = 0; ct < 2; ++ct)
                        for (int hl = 0; hl < 2; ++hl)
                            for (int gq = 0; gq < 4; ++gq)
                                for (int r = 0; r < 16; ++r)
                                    for (int j = 0; j < 8; ++j)
                                        wfrag[o++] = wpack_h[(long)hl * CP * K + (long)(nt * 32 + ct * 16 + r) * K + order[t] * ctot + ch * 32 + gq * 8 + j];
    } else if (g.tap_ok) {
        // [step][nt32][hi|lo][lane = h*32 + r][8], k = step*16 + h*8 + j, zero past the last unit
        const int kreal = taps * g.c0k;
        long o = 0;
        for (int st = 0; st < g.nsteps; ++st)
            for (int nt = 0; nt < nt32; ++nt)
                for (int hl = 0; hl < 2; ++hl)
                    for (int h = 0; h < 2; ++h)
                        for (int r = 0; r < 32; ++r)
                            for (int j = 0; j < 8; ++j) {
                                const int k = st * 16 + h * 8 + j;
                                wfrag[o++] = k < kreal ? wpack_h[(long)hl * CP * K + (long)(nt * 32 + r) * K + k] : (uint16_t)0;
                            }
    } else {
        set_error("pack_conv_weights: this layer has no fragment-order copy (see fusg_pack_conv_sizes.wfrag_order)");
        return FUSG_ERR_INVALID;
    }
    return FUSG_OK;
}

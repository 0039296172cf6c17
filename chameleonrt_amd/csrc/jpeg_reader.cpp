// jpeg_reader.cpp — JPEG texture decoding with the ARITHMETIC of the reference's decoder (harness).
//
// The reference reads every texture file through its vendored stb_image (`Image::Image(file, ...)`, util/material.cpp:5-17:
// stbi_set_flip_vertically_on_load(1); stbi_load(..., 4)). Two JPEG decoders agree on the entropy-decoded coefficients but
// not on the pixels: the inverse DCT's rounding, the chroma up-sampling filter and the fixed-point YCbCr -> RGB conversion are
// implementation choices (Pillow / libjpeg-turbo differs from stb_image by up to 2/255 on a few per cent of the samples). So
// those three steps are restated here from util/stb_image.h, value for value:
//   * inverse DCT: two passes of the 12-bit fixed-point "islow" butterfly, + 512 >> 10 after the columns,
//     + 65536 + (128 << 17) >> 17 after the rows, clamped            (stb_image.h:2213-2331, stbi__idct_block)
//   * chroma up-sampling: nearest (1x), (3a + b + 2) >> 2 (2x in one axis), (3 * t0 + t1 + 8) >> 4 of vertically pre-mixed
//     t = 3 * near + far (2x2), pixel replication otherwise         (stb_image.h:3229-3300, 3421-3432)
//   * YCbCr -> RGB in 20-bit fixed point with the green cross term masked to 16 bits (stb_image.h:3434-3459)
//   * which rows feed the vertical filter (line0 / line1 / ystep)   (stb_image.h:3664-3704)
// The entropy decoder (Huffman tables, baseline and progressive scans, restart intervals, DQT / DHT / DRI / SOF0-2 / APP0 / APP14)
// follows the JPEG specification (ITU T.81) like any other; where stb_image takes a liberty that shows in the pixels --
// coefficients are `short`s, products wrap -- it is kept. The SIMD kernels stb_image uses on x86 are, by its own comments,
// bit-identical to the scalar ones restated here (stb_image.h:2334-2336, 3434-3435).
// tests/test_texture_decoders.py holds this to the reference's decoder compiled in place (oracle/_ref/libref_scene.so).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/crt_scene_io.h"

namespace {

struct Fail : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// zig-zag position -> row-major position in the 8x8 block (ITU T.81 figure A.6)
const uint8_t DEZIGZAG[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                   6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable {
    // canonical code of ITU T.81 annex C: for code length L, codes [first[L], first[L] + count[L]) map to values[offset[L] ...]
    int count[17] = {0};
    int first[18] = {0};
    int offset[17] = {0};
    uint8_t values[256] = {0};
    bool defined = false;
    void build(const int sizes[16])
    {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            count[len] = sizes[len - 1];
            first[len] = code;
            offset[len] = k;
            code += count[len];
            if (count[len] && code - 1 >= (1 << len)) {
                throw Fail("bad code lengths");
            }
            k += count[len];
            code <<= 1;
        }
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
    int dc_pred = 0;
    int x = 0, y = 0;   // effective size in samples
    int w2 = 0, h2 = 0; // allocated plane: whole MCUs
    std::vector<uint8_t> plane;
    std::vector<int16_t> coeff; // progressive: 64 per block, blocks row-major, coeff_w per row
    int coeff_w = 0;
};

class Jpeg {
public:
    Jpeg(const uint8_t *p, size_t n) : cur(p), end(p + n) {}
    std::vector<uint8_t> decode_rgba(int &width, int &height)
    {
        read_headers_and_scans();
        if (progressive) {
            finish_progressive();
        }
        width = img_x;
        height = img_y;
        return to_rgba();
    }

private:
    const uint8_t *cur, *end;
    // frame
    int img_x = 0, img_y = 0, n_comp = 0;
    bool progressive = false, jfif = false;
    int adobe_transform = -1, rgb_ids = 0;
    int h_max = 1, v_max = 1, mcu_x = 0, mcu_y = 0;
    Component comp[4];
    uint16_t dequant[4][64] = {{0}};
    HuffTable dc_tab[4], ac_tab[4];
    int restart_interval = 0;
    // scan
    int scan_n = 0, order[4] = {0, 0, 0, 0};
    int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0, todo = 0;
    // entropy-coded bit reader: bits are taken from the top of `acc`; a marker ends the supply (zeros follow)
    uint32_t acc = 0;
    int n_bits = 0;
    int pending_marker = 0xff;
    bool no_more = false;

    int get8() { return cur < end ? *cur++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }
    bool at_eof() const { return cur >= end; }
    void skip(int n) { cur = (end - cur < n) ? end : cur + n; }

    void fill()
    {
        do {
            const unsigned b = no_more ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) {
                    c = get8(); // fill bytes
                }
                if (c != 0) {
                    pending_marker = c;
                    no_more = true;
                    return;
                }
            }
            acc |= b << (24 - n_bits);
            n_bits += 8;
        } while (n_bits <= 24);
    }
    int get_bits(int n)
    {
        if (n == 0) {
            return 0;
        }
        if (n_bits < n) {
            fill();
        }
        const uint32_t k = acc >> (32 - n);
        acc <<= n;
        n_bits -= n;
        return (int)k;
    }
    int get_bit() { return get_bits(1); }
    // ITU T.81 F.2.2.1 EXTEND of a RECEIVEd n-bit value
    int receive_extend(int n)
    {
        if (n == 0) {
            return 0;
        }
        if (n_bits < n) {
            fill();
        }
        const int sign_is_one = (int)(acc >> 31);
        const int k = get_bits(n);
        return sign_is_one ? k : k + (int)((~0u << n) + 1u);
    }
    int huff_decode(const HuffTable &t)
    {
        if (!t.defined) {
            throw Fail("undefined Huffman table");
        }
        if (n_bits < 16) {
            fill();
        }
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (int)(acc >> (32 - len));
            if (t.count[len] && code >= t.first[len] && code < t.first[len] + t.count[len]) {
                if (len > n_bits) {
                    return -1;
                }
                acc <<= len;
                n_bits -= len;
                return t.values[t.offset[len] + code - t.first[len]];
            }
        }
        n_bits -= 16;
        return -1;
    }

    int next_marker()
    {
        if (pending_marker != 0xff) {
            const int m = pending_marker;
            pending_marker = 0xff;
            return m;
        }
        int x = get8();
        if (x != 0xff) {
            return 0xff;
        }
        while (x == 0xff) {
            x = get8();
        }
        return x;
    }

    void process_marker(int m)
    {
        if (m == 0xff) {
            throw Fail("expected marker");
        }
        if (m == 0xDD) { // DRI
            if (get16() != 4) {
                throw Fail("bad DRI len");
            }
            restart_interval = get16();
            return;
        }
        if (m == 0xDB) { // DQT
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), p = q >> 4, t = q & 15;
                if ((p != 0 && p != 1) || t > 3) {
                    throw Fail("bad DQT");
                }
                for (int i = 0; i < 64; ++i) {
                    dequant[t][DEZIGZAG[i]] = (uint16_t)(p ? get16() : get8());
                }
                L -= p ? 129 : 65;
            }
            if (L != 0) {
                throw Fail("bad DQT len");
            }
            return;
        }
        if (m == 0xC4) { // DHT
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) {
                    throw Fail("bad DHT header");
                }
                int sizes[16], n = 0;
                for (int i = 0; i < 16; ++i) {
                    sizes[i] = get8();
                    n += sizes[i];
                }
                if (n > 256) {
                    throw Fail("bad DHT");
                }
                HuffTable &t = tc == 0 ? dc_tab[th] : ac_tab[th];
                t.build(sizes);
                for (int i = 0; i < n; ++i) {
                    t.values[i] = (uint8_t)get8();
                }
                L -= 17 + n;
            }
            if (L != 0) {
                throw Fail("bad DHT len");
            }
            return;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) { // APPn / COM
            int L = get16();
            if (L < 2) {
                throw Fail("bad APP / COM len");
            }
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const char tag[5] = {'J', 'F', 'I', 'F', '\0'};
                bool ok = true;
                for (int i = 0; i < 5; ++i) {
                    ok = (get8() == tag[i]) && ok;
                }
                L -= 5;
                jfif = jfif || ok;
            } else if (m == 0xEE && L >= 12) {
                static const char tag[6] = {'A', 'd', 'o', 'b', 'e', '\0'};
                bool ok = true;
                for (int i = 0; i < 6; ++i) {
                    ok = (get8() == tag[i]) && ok;
                }
                L -= 6;
                if (ok) {
                    get8();
                    get16();
                    get16();
                    adobe_transform = get8();
                    L -= 6;
                }
            }
            skip(L);
            return;
        }
        throw Fail("unknown marker");
    }

    void frame_header()
    {
        const int Lf = get16();
        if (Lf < 11 || get8() != 8) {
            throw Fail("unsupported SOF");
        }
        img_y = get16();
        img_x = get16();
        n_comp = get8();
        if (img_x == 0 || img_y == 0 || (n_comp != 1 && n_comp != 3 && n_comp != 4) || Lf != 8 + 3 * n_comp) {
            throw Fail("bad SOF");
        }
        if ((uint64_t)img_x * (uint64_t)img_y > (1ull << 28)) {
            throw Fail("image too large");
        }
        for (int i = 0; i < n_comp; ++i) {
            static const char rgb[3] = {'R', 'G', 'B'};
            comp[i].id = get8();
            if (n_comp == 3 && comp[i].id == rgb[i]) {
                ++rgb_ids;
            }
            const int q = get8();
            comp[i].h = q >> 4;
            comp[i].v = q & 15;
            comp[i].tq = get8();
            if (!comp[i].h || comp[i].h > 4 || !comp[i].v || comp[i].v > 4 || comp[i].tq > 3) {
                throw Fail("bad component");
            }
            h_max = std::max(h_max, comp[i].h);
            v_max = std::max(v_max, comp[i].v);
        }
        // every component's sampling factor must divide the largest one: to_rgba up-samples by the integer ratio h_max / h, and
        // with H = 3, 2, 1 that ratio would read past a plane only mcu_x * h * 8 wide. A DELIBERATE DIVERGENCE for malformed files:
        // the reference's bundled stb_image (v2.23, util/stb_image.h) has no such check and decodes them (reading out of bounds);
        // later stb_image releases added it. Refusing is the safe choice; it is not parity for these files.
        for (int i = 0; i < n_comp; ++i) {
            if (h_max % comp[i].h != 0 || v_max % comp[i].v != 0) {
                throw Fail("bad H / V: sampling factors that do not divide the largest");
            }
        }
        mcu_x = (img_x + h_max * 8 - 1) / (h_max * 8);
        mcu_y = (img_y + v_max * 8 - 1) / (v_max * 8);
        for (int i = 0; i < n_comp; ++i) {
            Component &c = comp[i];
            c.x = (img_x * c.h + h_max - 1) / h_max;
            c.y = (img_y * c.v + v_max - 1) / v_max;
            c.w2 = mcu_x * c.h * 8;
            c.h2 = mcu_y * c.v * 8;
            c.plane.assign((size_t)c.w2 * c.h2, 0);
            if (progressive) {
                c.coeff_w = c.w2 / 8;
                c.coeff.assign((size_t)c.w2 * c.h2, 0);
            }
        }
    }

    void scan_header()
    {
        const int Ls = get16();
        scan_n = get8();
        if (scan_n < 1 || scan_n > 4 || scan_n > n_comp || Ls != 6 + 2 * scan_n) {
            throw Fail("bad SOS");
        }
        for (int i = 0; i < scan_n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < n_comp && comp[which].id != id) {
                ++which;
            }
            if (which == n_comp || (q >> 4) > 3 || (q & 15) > 3) {
                throw Fail("bad SOS component");
            }
            comp[which].hd = q >> 4;
            comp[which].ha = q & 15;
            order[i] = which;
        }
        spec_start = get8();
        spec_end = get8();
        const int aa = get8();
        succ_high = aa >> 4;
        succ_low = aa & 15;
        if (progressive) {
            if (spec_start > 63 || spec_end > 63 || spec_start > spec_end || succ_high > 13 || succ_low > 13) {
                throw Fail("bad SOS");
            }
        } else {
            if (spec_start != 0 || succ_high != 0 || succ_low != 0) {
                throw Fail("bad SOS");
            }
            spec_end = 63;
        }
    }

    void reset_entropy()
    {
        n_bits = 0;
        acc = 0;
        no_more = false;
        for (Component &c : comp) {
            c.dc_pred = 0;
        }
        pending_marker = 0xff;
        todo = restart_interval ? restart_interval : 0x7fffffff;
        eob_run = 0;
    }
    // after a data unit / MCU: count the restart interval down; false = the scan ends here (no RSTn where one was due)
    bool interval_step()
    {
        if (--todo <= 0) {
            if (n_bits < 24) {
                fill();
            }
            if (!(pending_marker >= 0xd0 && pending_marker <= 0xd7)) {
                return false;
            }
            reset_entropy();
        }
        return true;
    }

    // baseline: one 8x8 block, dequantised on the fly; the products are stored as 16-bit values (stb_image.h:2021-2071)
    void decode_block(int16_t data[64], Component &c)
    {
        const int t = huff_decode(dc_tab[c.hd]);
        if (t < 0 || t > 15) { // (a DC category is a bit count: more than 15 would shift by >= 32 in receive_extend; the reference's stb_image v2.23 does not check and has undefined behaviour there -- refusing the file is deliberate, not parity)
            throw Fail("bad huffman code");
        }
        std::memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? receive_extend(t) : 0;
        const int dc = c.dc_pred + diff;
        c.dc_pred = dc;
        const uint16_t *dq = dequant[c.tq];
        data[0] = (int16_t)(dc * dq[0]);
        int k = 1;
        do {
            const int rs = huff_decode(ac_tab[c.ha]);
            if (rs < 0) {
                throw Fail("bad huffman code");
            }
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) {
                    break;
                }
                k += 16;
            } else {
                k += r;
                const unsigned zig = DEZIGZAG[std::min(k, 78)]; // (a corrupt run may point past 63: the table is padded for that)
                ++k;
                data[zig] = (int16_t)(receive_extend(s) * dq[zig]);
            }
        } while (k < 64);
    }
    void prog_dc(int16_t *data, Component &c)
    {
        if (spec_end != 0) {
            throw Fail("can't merge dc and ac");
        }
        if (succ_high == 0) {
            std::memset(data, 0, 64 * sizeof(int16_t));
            const int t = huff_decode(dc_tab[c.hd]);
            if (t < 0 || t > 15) {
                throw Fail("bad huffman code");
            }
            const int diff = t ? receive_extend(t) : 0;
            const int dc = c.dc_pred + diff;
            c.dc_pred = dc;
            data[0] = (int16_t)(dc * (1 << succ_low)); // (a product, not a shift: dc may be negative)
        } else if (get_bit()) {
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succ_low));
        }
    }
    static void refine(int16_t *p, int16_t bit, Jpeg &j)
    {
        if (*p != 0 && j.get_bit() && (*p & bit) == 0) {
            *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
        }
    }
    void prog_ac(int16_t *data, Component &c)
    {
        if (spec_start == 0) {
            throw Fail("can't merge dc and ac");
        }
        const HuffTable &tab = ac_tab[c.ha];
        if (succ_high == 0) {
            if (eob_run) {
                --eob_run;
                return;
            }
            int k = spec_start;
            do {
                const int rs = huff_decode(tab);
                if (rs < 0) {
                    throw Fail("bad huffman code");
                }
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eob_run = (1 << r);
                        if (r) {
                            eob_run += get_bits(r);
                        }
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const unsigned zig = DEZIGZAG[std::min(k, 78)];
                    ++k;
                    data[zig] = (int16_t)(receive_extend(s) * (1 << succ_low));
                }
            } while (k <= spec_end);
            return;
        }
        const int16_t bit = (int16_t)(1 << succ_low);
        if (eob_run) {
            --eob_run;
            for (int k = spec_start; k <= spec_end; ++k) {
                refine(&data[DEZIGZAG[k]], bit, *this);
            }
            return;
        }
        int k = spec_start;
        do {
            const int rs = huff_decode(tab);
            if (rs < 0) {
                throw Fail("bad huffman code");
            }
            int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) {
                        eob_run += get_bits(r);
                    }
                    r = 64; // to the end of the band
                }
            } else {
                if (s != 1) {
                    throw Fail("bad huffman code");
                }
                s = get_bit() ? bit : -bit;
            }
            while (k <= spec_end) {
                int16_t *p = &data[DEZIGZAG[k++]];
                if (*p != 0) {
                    refine(p, bit, *this);
                } else {
                    if (r == 0) {
                        *p = (int16_t)s;
                        break;
                    }
                    --r;
                }
            }
        } while (k <= spec_end);
    }

    // stb_image.h:2213-2331 (derived, as it says, from jidctint's islow): constants scaled by 4096 and rounded the way its macro rounds
    static constexpr int fx(double x) { return (int)(x * 4096 + 0.5); }
    struct Pass {
        int x0, x1, x2, x3, t0, t1, t2, t3;
    };
    static Pass butterfly(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7)
    {
        int t0, t1, t2, t3, p1, p2, p3, p4, p5;
        p2 = s2;
        p3 = s6;
        p1 = (p2 + p3) * fx(0.5411961f);
        t2 = p1 + p3 * fx(-1.847759065f);
        t3 = p1 + p2 * fx(0.765366865f);
        p2 = s0;
        p3 = s4;
        t0 = (p2 + p3) * 4096;
        t1 = (p2 - p3) * 4096;
        Pass r;
        r.x0 = t0 + t3;
        r.x3 = t0 - t3;
        r.x1 = t1 + t2;
        r.x2 = t1 - t2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        p5 = (p3 + p4) * fx(1.175875602f);
        t0 = t0 * fx(0.298631336f);
        t1 = t1 * fx(2.053119869f);
        t2 = t2 * fx(3.072711026f);
        t3 = t3 * fx(1.501321110f);
        p1 = p5 + p1 * fx(-0.899976223f);
        p2 = p5 + p2 * fx(-2.562915447f);
        p3 = p3 * fx(-1.961570560f);
        p4 = p4 * fx(-0.390180644f);
        r.t3 = t3 + p1 + p4;
        r.t2 = t2 + p2 + p3;
        r.t1 = t1 + p2 + p4;
        r.t0 = t0 + p1 + p3;
        return r;
    }
    static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }
    static void idct(uint8_t *out, int stride, const int16_t d[64])
    {
        int val[64];
        for (int i = 0; i < 8; ++i) { // columns
            const int16_t *c = d + i;
            int *v = val + i;
            if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
                const int dcterm = c[0] * 4;
                v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
            } else {
                Pass p = butterfly(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
                p.x0 += 512;
                p.x1 += 512;
                p.x2 += 512;
                p.x3 += 512;
                v[0] = (p.x0 + p.t3) >> 10;
                v[56] = (p.x0 - p.t3) >> 10;
                v[8] = (p.x1 + p.t2) >> 10;
                v[48] = (p.x1 - p.t2) >> 10;
                v[16] = (p.x2 + p.t1) >> 10;
                v[40] = (p.x2 - p.t1) >> 10;
                v[24] = (p.x3 + p.t0) >> 10;
                v[32] = (p.x3 - p.t0) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i) { // rows
            const int *v = val + 8 * i;
            uint8_t *o = out + (size_t)stride * i;
            Pass p = butterfly(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            const int bias = 65536 + (128 << 17);
            p.x0 += bias;
            p.x1 += bias;
            p.x2 += bias;
            p.x3 += bias;
            o[0] = clamp8((p.x0 + p.t3) >> 17);
            o[7] = clamp8((p.x0 - p.t3) >> 17);
            o[1] = clamp8((p.x1 + p.t2) >> 17);
            o[6] = clamp8((p.x1 - p.t2) >> 17);
            o[2] = clamp8((p.x2 + p.t1) >> 17);
            o[5] = clamp8((p.x2 - p.t1) >> 17);
            o[3] = clamp8((p.x3 + p.t0) >> 17);
            o[4] = clamp8((p.x3 - p.t0) >> 17);
        }
    }

    // the data units of one scan, in the order of ITU T.81 A.2: a single component block by block over its own extent,
    // several components MCU by MCU (stb_image.h:2753-2875)
    void entropy_coded_data()
    {
        reset_entropy();
        int16_t block[64];
        if (scan_n == 1) {
            Component &c = comp[order[0]];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j) {
                for (int i = 0; i < w; ++i) {
                    if (!progressive) {
                        decode_block(block, c);
                        idct(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                    } else {
                        int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                        if (spec_start == 0) {
                            prog_dc(data, c);
                        } else {
                            prog_ac(data, c);
                        }
                    }
                    if (!interval_step()) {
                        return;
                    }
                }
            }
            return;
        }
        for (int j = 0; j < mcu_y; ++j) {
            for (int i = 0; i < mcu_x; ++i) {
                for (int k = 0; k < scan_n; ++k) {
                    Component &c = comp[order[k]];
                    for (int y = 0; y < c.v; ++y) {
                        for (int x = 0; x < c.h; ++x) {
                            const int bx = i * c.h + x, by = j * c.v + y;
                            if (!progressive) {
                                decode_block(block, c);
                                idct(c.plane.data() + (size_t)c.w2 * by * 8 + bx * 8, c.w2, block);
                            } else {
                                prog_dc(c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.coeff_w), c);
                            }
                        }
                    }
                }
                if (!interval_step()) {
                    return;
                }
            }
        }
    }

    void finish_progressive()
    {
        for (int n = 0; n < n_comp; ++n) {
            Component &c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j) {
                for (int i = 0; i < w; ++i) {
                    int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeff_w);
                    for (int k = 0; k < 64; ++k) {
                        data[k] = (int16_t)(data[k] * dequant[c.tq][k]); // (16-bit product, like the baseline path)
                    }
                    idct(c.plane.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
                }
            }
        }
    }

    void read_headers_and_scans()
    {
        if (next_marker() != 0xd8) {
            throw Fail("no SOI");
        }
        int m = next_marker();
        while (!(m == 0xc0 || m == 0xc1 || m == 0xc2)) {
            process_marker(m);
            m = next_marker();
            while (m == 0xff) {
                if (at_eof()) {
                    throw Fail("no SOF");
                }
                m = next_marker();
            }
        }
        progressive = m == 0xc2;
        frame_header();
        m = next_marker();
        while (m != 0xd9) {
            if (m == 0xda) {
                scan_header();
                entropy_coded_data();
                if (pending_marker == 0xff) { // stray bytes after the entropy-coded segment: look for the next marker
                    while (!at_eof()) {
                        if (get8() == 255) {
                            pending_marker = get8();
                            break;
                        }
                    }
                }
            } else if (m == 0xdc) { // DNL
                const int Ld = get16(), NL = get16();
                if (Ld != 4 || NL != img_y) {
                    throw Fail("bad DNL");
                }
            } else {
                process_marker(m);
            }
            m = next_marker();
            if (m == 0xff && at_eof()) {
                throw Fail("no EOI");
            }
        }
    }

    // ---- planes -> RGBA: stb_image.h:3640-3790 with req_comp = 4 (the reference asks for four channels) -------------------
    static void up_v2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w)
    {
        for (int i = 0; i < w; ++i) {
            out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        }
    }
    static void up_h2(uint8_t *out, const uint8_t *in, int w)
    {
        if (w == 1) {
            out[0] = out[1] = in[0];
            return;
        }
        out[0] = in[0];
        out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; ++i) {
            const int n = 3 * in[i] + 2;
            out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
            out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
        out[i * 2 + 1] = in[w - 1];
    }
    static void up_hv2(uint8_t *out, const uint8_t *near, const uint8_t *far, int w)
    {
        if (w == 1) {
            out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2);
            return;
        }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; ++i) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    }
    static constexpr int f2fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
    static void ycbcr_row(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int count)
    {
        for (int i = 0; i < count; ++i) {
            const int y_fixed = (y[i] << 20) + (1 << 19);
            const int cr = pcr[i] - 128, cb = pcb[i] - 128;
            int r = y_fixed + cr * f2fixed(1.40200f);
            int g = y_fixed + (cr * -f2fixed(0.71414f)) + (int)(((unsigned)(cb * -f2fixed(0.34414f))) & 0xffff0000u);
            int b = y_fixed + cb * f2fixed(1.77200f);
            r >>= 20;
            g >>= 20;
            b >>= 20;
            out[4 * i + 0] = clamp8(r);
            out[4 * i + 1] = clamp8(g);
            out[4 * i + 2] = clamp8(b);
            out[4 * i + 3] = 255;
        }
    }
    static uint8_t blinn(uint8_t x, uint8_t y)
    {
        const unsigned t = (unsigned)x * y + 128;
        return (uint8_t)((t + (t >> 8)) >> 8);
    }

    std::vector<uint8_t> to_rgba()
    {
        const bool is_rgb = n_comp == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
        struct Up {
            int hs, vs, w_lores, ystep, ypos;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buf;
        } up[4];
        for (int k = 0; k < n_comp; ++k) {
            Up &r = up[k];
            r.hs = h_max / comp[k].h;
            r.vs = v_max / comp[k].v;
            r.ystep = r.vs >> 1;
            r.w_lores = (img_x + r.hs - 1) / r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = comp[k].plane.data();
            r.buf.assign((size_t)img_x + 8 + 2 * (size_t)r.w_lores * (size_t)std::max(1, r.hs), 0);
        }
        std::vector<uint8_t> out((size_t)img_x * img_y * 4);
        const uint8_t *row[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int j = 0; j < img_y; ++j) {
            uint8_t *o = out.data() + (size_t)4 * img_x * j;
            for (int k = 0; k < n_comp; ++k) {
                Up &r = up[k];
                const bool y_bot = r.ystep >= (r.vs >> 1);
                const uint8_t *near = y_bot ? r.line1 : r.line0, *far = y_bot ? r.line0 : r.line1;
                if (r.hs == 1 && r.vs == 1) {
                    row[k] = near;
                } else {
                    uint8_t *b = r.buf.data();
                    if (r.hs == 1 && r.vs == 2) {
                        up_v2(b, near, far, r.w_lores);
                    } else if (r.hs == 2 && r.vs == 1) {
                        up_h2(b, near, r.w_lores);
                    } else if (r.hs == 2 && r.vs == 2) {
                        up_hv2(b, near, far, r.w_lores);
                    } else {
                        for (int i = 0; i < r.w_lores; ++i) {
                            for (int s = 0; s < r.hs; ++s) {
                                b[i * r.hs + s] = near[i];
                            }
                        }
                    }
                    row[k] = b;
                }
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < comp[k].y) {
                        r.line1 += comp[k].w2;
                    }
                }
            }
            if (n_comp == 3) {
                if (is_rgb) {
                    for (int i = 0; i < img_x; ++i) {
                        o[4 * i] = row[0][i];
                        o[4 * i + 1] = row[1][i];
                        o[4 * i + 2] = row[2][i];
                        o[4 * i + 3] = 255;
                    }
                } else {
                    ycbcr_row(o, row[0], row[1], row[2], img_x);
                }
            } else if (n_comp == 4) {
                if (adobe_transform == 0) { // CMYK
                    for (int i = 0; i < img_x; ++i) {
                        const uint8_t m = row[3][i];
                        o[4 * i] = blinn(row[0][i], m);
                        o[4 * i + 1] = blinn(row[1][i], m);
                        o[4 * i + 2] = blinn(row[2][i], m);
                        o[4 * i + 3] = 255;
                    }
                } else {
                    ycbcr_row(o, row[0], row[1], row[2], img_x);
                    if (adobe_transform == 2) { // YCCK
                        for (int i = 0; i < img_x; ++i) {
                            const uint8_t m = row[3][i];
                            o[4 * i] = blinn((uint8_t)(255 - o[4 * i]), m);
                            o[4 * i + 1] = blinn((uint8_t)(255 - o[4 * i + 1]), m);
                            o[4 * i + 2] = blinn((uint8_t)(255 - o[4 * i + 2]), m);
                        }
                    }
                }
            } else {
                for (int i = 0; i < img_x; ++i) {
                    o[4 * i] = o[4 * i + 1] = o[4 * i + 2] = row[0][i];
                    o[4 * i + 3] = 255;
                }
            }
        }
        return out;
    }
};

thread_local std::string g_image_error;

} // namespace

extern "C" {

int crt_image_decode_jpeg(const uint8_t *bytes, uint64_t n_bytes, int32_t *width, int32_t *height, uint8_t **rgba)
{
    if (!bytes || !width || !height || !rgba) {
        g_image_error = "null argument";
        return -1;
    }
    try {
        int w = 0, h = 0;
        Jpeg j(bytes, (size_t)n_bytes);
        std::vector<uint8_t> px = j.decode_rgba(w, h);
        uint8_t *out = new uint8_t[px.size()];
        std::memcpy(out, px.data(), px.size());
        *width = w;
        *height = h;
        *rgba = out;
        return 0;
    } catch (const std::exception &e) {
        g_image_error = e.what();
        return -1;
    }
}

void crt_image_free(uint8_t *rgba) { delete[] rgba; }
const char *crt_image_error(void) { return g_image_error.c_str(); }

} // extern "C"

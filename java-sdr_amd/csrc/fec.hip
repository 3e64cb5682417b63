// fec.hip -- AO-40 FEC on the GPU: FECDecoder.FECDecode (FECDecoder.java:703-852) and the re-encoder
// encode_FEC40 (:538-688).  Integer only, bit-exact by construction.
//
// MI355X mapping: ONE WAVEFRONT PER 5200-SYMBOL BLOCK.
//   * Viterbi K=7 r=1/2 has 64 states = the 64 lanes of a wave.  Lane s keeps the path metric of state s
//     in a VGPR; the two predecessors (s>>1, (s>>1)+32) arrive by cross-lane reads; the 64 decisions of a
//     step are one __ballot -- the same 64 bits the reference packs into its two `long pp[]` words per step
//     (FECDecoder.java:229-255) -- kept in LDS for the chain-back.
//   * RS(255,223) x2: the 2 x 32 syndromes are the 64 lanes (Horner over the 160 non-padding columns); when a
//     syndrome is non-zero the two code words are corrected on the two half-waves, 32 lanes each: Berlekamp-
//     Massey with the polynomials spread over the lanes, Chien search as ballots over the 255 field elements,
//     Forney with one root per lane (rs_correct_halfwave; same results as :387-511).
//   * re-encode: the two RS parity LFSRs run on the two half-waves (lane = register position), the
//     convolutional encoder + interleaver is parallel over the 2566 bits.
#include "bpsk_fec.h"
#include <vector>

namespace jsdr {

enum {
    NN = 255, KK = 223, NROOTS = 32, FCR = 112, PRIM = 11, IPRIM = 116, A0 = 255,
    RSPAD = 95, NBITS = 2566, ROWS = 80, COLUMNS = 65, SYMPBLOCK = 5200
};

struct FecTables {
    unsigned char alpha_to[256];
    unsigned char index_of[256];
    unsigned char scrambler[320];
    unsigned char rs_coef[32];  // index-form generator coefficient applied to register position q (q=1..31), [0] unused
    unsigned char sync[65];     // sync vector bits (1/0), FECDecoder.java:600-605
    short mettab[2][256];
};

__constant__ FecTables c_fec;

// FECDecoder.java:67-83 mettab[0] (sent symbol 0); row [1] is its mirror except two entries (:84).
static const short h_mettab0[256] = {
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 18, 18, 18, 18, 18, 18, 17, 17, 17, 16, 16, 16, 15, 15, 14, 14, 13, 13,
    12, 11, 10, 10, 9, 8, 7, 6, 5, 3, 2, 1, -1, -2, -4, -5, -7, -9, -11, -13, -15, -17, -19, -21, -23, -25, -28, -30,
    -32, -35, -37, -40, -42, -45, -47, -50, -52, -55, -58, -60, -63, -66, -68, -71, -74, -77, -79, -82, -85, -88, -90,
    -93, -96, -99, -102, -104, -107, -110, -113, -116, -119, -121, -124, -127, -130, -133, -136, -138, -141, -144, -147,
    -150, -153, -155, -158, -161, -164, -167, -170, -172, -175, -178, -181, -184, -187, -190, -192, -195, -198, -201,
    -204, -207, -210, -212, -215, -218, -221, -224, -227, -229, -232, -235, -238, -241, -244, -247, -249, -252, -255,
    -258, -261, -264, -267, -269, -272, -275, -278, -281, -284, -286, -289, -292, -295, -298, -301, -304, -306, -309,
    -312, -315, -318, -320, -324, -326, -329, -332, -335, -337, -341, -372};

static bool g_tables_ready[64] = {false};

static int upload_tables()
{
    int dev = 0;
    JSDR_HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && g_tables_ready[dev]) return JSDR_OK;
    FecTables t;
    memset(&t, 0, sizeof(t));
    // GF(256), field polynomial 0x187 (FECDecoder.java:145-181)
    int x = 1;
    for (int i = 0; i < 255; i++) {
        t.alpha_to[i] = (unsigned char)x;
        t.index_of[x] = (unsigned char)i;
        x <<= 1;
        if (x & 0x100) x ^= 0x187;
    }
    t.alpha_to[255] = 0;
    t.index_of[0] = A0;
    // CCSDS randomiser x^8+x^7+x^5+x^3+1, all-ones start (:118-139)
    unsigned sr = 0xff;
    for (int i = 0; i < 320; i++) {
        int byte = 0;
        for (int b = 0; b < 8; b++) {
            byte = (byte << 1) | (int)(sr & 1u);
            unsigned fb = (sr ^ (sr >> 3) ^ (sr >> 5) ^ (sr >> 7)) & 1u;
            sr = (sr >> 1) | (fb << 7);
        }
        t.scrambler[i] = (unsigned char)byte;
    }
    // RS generator prod_{i<32}(x - alpha^{PRIM(FCR+i)}), index form; palindromic (:544-546, :634-640)
    int g[NROOTS + 1];
    memset(g, 0, sizeof(g));
    g[0] = 1;
    for (int i = 0; i < NROOTS; i++) {
        int root = ((FCR + i) * PRIM) % 255;
        g[i + 1] = 1;
        for (int j = i; j > 0; j--) {
            if (g[j] != 0)
                g[j] = g[j - 1] ^ t.alpha_to[(t.index_of[g[j]] + root) % 255];
            else
                g[j] = g[j - 1];
        }
        g[0] = t.alpha_to[(t.index_of[g[0]] + root) % 255];
    }
    for (int q = 1; q < 32; q++) t.rs_coef[q] = t.index_of[g[q]];
    // sync LFSR (:600-605)
    int s7 = 0x7f;
    for (int i = 0; i < 65; i++) {
        t.sync[i] = (s7 & 64) ? 1 : 0;
        int v = s7 & 0x48;
        v ^= v >> 4;
        v ^= v >> 2;
        v ^= v >> 1;
        s7 = ((s7 << 1) | (v & 1)) & 0xffff;
    }
    for (int i = 0; i < 256; i++) {
        t.mettab[0][i] = h_mettab0[i];
        t.mettab[1][i] = h_mettab0[255 - i];
    }
    t.mettab[1][2] = -338;
    t.mettab[1][8] = -321;
    JSDR_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_fec), &t, sizeof(t)));
    if (dev >= 0 && dev < 64) g_tables_ready[dev] = true;
    return JSDR_OK;
}

int fec_prepare() { return upload_tables(); }

// ----------------------------------------------------------------------------------------------
// LDS work area of one wave
// DECW = NBITS+2: the decision words live in LDS (31.5 KB per block).  DECW = DECW_WORK: they live in a global
// scratch area and LDS keeps only the work area behind them (16.2 KB per block): the demodulator's FEC kernel runs
// under the throughput kernels of the next call, where every LDS byte it holds is a workgroup of theirs that
// cannot start.
enum { DECW_LDS = NBITS + 2, DECW_WORK = (SYMPBLOCK + 7) / 8 + 6, DEC_SCRATCH_WORDS = ((NBITS + 2 + 63) / 64) * 64 };
template <int DECW>
struct FecLdsT {
    // The demodulator's instance (decision words in global scratch) decodes HARD decisions: one bit per symbol instead
    // of a soft byte, and a shorter branch-metric chunk -- 8.6 KB of LDS per block instead of 16.2, so that a CU holds
    // nineteen of these one-wave workgroups instead of ten (the kernel is occupancy x latency bound).
    static constexpr bool HARD = DECW != NBITS + 2;
    static constexpr int METS = HARD ? 128 : 512;
    unsigned char raw[HARD ? ((SYMPBLOCK + 63) / 64) * 8 : SYMPBLOCK];  // soft symbols in (HARD: bit i of the array = symbol i is a 1)
    __device__ __forceinline__ int sym(int i) const
    {
        if constexpr (HARD) return ((reinterpret_cast<const unsigned *>(raw)[i >> 5] >> (i & 31)) & 1u) ? 0xc0 : 0x40;
        else return raw[i];
    }
    unsigned long long dec[DECW];       // decisions per step (== the reference's pp[2k], pp[2k+1]); once the
                                        // chain-back is done the same bytes hold the RS work arrays, then the
                                        // re-encoded symbols (fec_enc())
    short mets[METS][4];                // branch metrics of the current chunk of trellis steps
    unsigned char alpha_to[256];
    unsigned char index_of[256];
    unsigned char vit[320];             // Viterbi output / scrambled byte stream
    unsigned char rs[2][256];           // RS code words (255 used)
    unsigned char data[256];            // decoded payload
    int misc[8];
};
typedef FecLdsT<DECW_LDS> FecLds;
template <int DECW>
__device__ __forceinline__ unsigned char *fec_enc(FecLdsT<DECW> &L) { return reinterpret_cast<unsigned char *>(&L.dec[0]); }

__device__ __forceinline__ int parity7(int v)
{
    return __popc(v) & 1;
}

__device__ __forceinline__ int gf_mod255(int x) { return x % 255; }

// encode_FEC40 (:677-688): data[256] (LDS) -> L.enc[5200]; 64 lanes cooperate.
template <int DECW>
__device__ __forceinline__ void fec_encode_wave(FecLdsT<DECW> &L, const unsigned char *data, int lane)
{
    unsigned char *enc = fec_enc(L);
    // ---- RS parity, two interleaved code words: lanes 0..31 block 0 (even bytes), 32..63 block 1 (odd)
    const int blk = lane >> 5, q = lane & 31;
    int reg = 0;
    const int coef = (q < 31) ? (int)c_fec.rs_coef[q + 1] : 0;  // generator coefficient feeding position q after the shift
    for (int n = 0; n < 128; n++) {
        int c = data[2 * n + blk];
        int r0 = __shfl(reg, blk * 32, 64);
        int fb = L.index_of[c ^ r0];           // :623
        int up = __shfl_down(reg, 1, 64);      // old RS_block[q+1]
        int nv;
        if (q < 31) {
            nv = up;
            if (fb != A0) nv ^= L.alpha_to[gf_mod255(fb + coef)];   // :634-645
        } else {
            nv = (fb != A0) ? (int)L.alpha_to[fb] : 0;              // :648-652
        }
        reg = nv;
    }
    // byte stream: 256 data + 64 parity (parity byte 256+2q+b = RS_block[b][q], :665), scrambled (:570)
    for (int i = lane; i < 256; i += 64) L.vit[i] = data[i] ^ c_fec.scrambler[i];
    L.vit[256 + 2 * q + blk] = (unsigned char)(reg ^ c_fec.scrambler[256 + 2 * q + blk]);
    for (int i = lane; i < SYMPBLOCK; i += 64) enc[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // sync vector in interleaver column 0 (:600-605)
    for (int i = lane; i < 65; i += 64)
        if (c_fec.sync[i]) enc[ROWS * i] = 1;
    // convolutional encoder + interleaver, parallel over the 2566 bits (:559-566, :549-556)
    for (int k = lane; k < NBITS; k += 64) {
        int sr = 0;
#pragma unroll
        for (int d = 6; d >= 0; d--) {
            int kk = k - d;
            int bit = 0;
            if (kk >= 0 && kk < 2560) bit = (L.vit[kk >> 3] >> (7 - (kk & 7))) & 1;
            sr = (sr << 1) | bit;
        }
        int a = parity7(sr & 0x4f);
        int b = 1 - parity7(sr & 0x6d);
        int bi = COLUMNS + 2 * k;
        if (a) enc[(bi % COLUMNS) * ROWS + bi / COLUMNS] = 1;
        bi++;
        if (b) enc[(bi % COLUMNS) * ROWS + bi / COLUMNS] = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------- RS correction
// Errors-only RS(255,223) correction of the block's two code words, ONE HALF-WAVE PER CODE WORD, after the
// syndromes (which are already lane-parallel).  Same results as decode_rs_8 (FECDecoder.java:387-511) -- the
// number of corrected symbols, or -1 when the locator's degree and its root count differ (:474) or an error
// value is undefined (:503) -- but organised for 32 lanes instead of one:
//   locator   : Berlekamp-Massey with the polynomials spread over the lanes.  Lane q keeps the coefficient
//               Lambda_{q+1} (Lambda_0 is 1 throughout) and B_q; a step is one GF multiply per lane, a 5-level
//               xor-reduction for the discrepancy, and neighbour shuffles for x.B(x) and B <- Lambda / Delta.
//   roots     : Chien search with the 255 field elements dealt over the lanes, 8 rounds; a ballot per round gives
//               the roots in increasing order, a prefix popcount their slot in the root list.
//   evaluator : Omega = S . Lambda mod x^32, one coefficient per lane.
//   values    : Forney with one root per lane: numerator, formal derivative, and the byte patched in place.
// The data-dependent trip counts (deg Lambda, deg Omega, root count) of the two code words differ, so both
// half-waves run the wave-uniform maximum under a per-lane guard.
#define FEC_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

struct RsWork {            // per code word, in LDS; index form (A0 = the zero element) unless noted
    short lam[NROOTS + 2]; // locator coefficients 0..32
    short omg[NROOTS];     // evaluator coefficients
    short root[NROOTS];    // exponents i with Lambda(alpha^i) = 0, ascending
    short loc[NROOTS];     // the code word positions they stand for
};
static_assert(sizeof(RsWork) <= 272 * sizeof(short), "RsWork must fit the per-code-word slice of the dead decision area");

__device__ __forceinline__ int xor_reduce32(int v)  // over the 32 lanes of a half-wave
{
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) v ^= __shfl_xor(v, off, 64);
    return v;
}

// cw: the code word of this half-wave (255 bytes, LDS); syn: its 32 syndromes, index form (LDS); W: its work area.
// q = lane & 31.  Returns the half-wave-uniform result.
__device__ __forceinline__ int rs_correct_halfwave(unsigned char *cw, const int *syn, const unsigned char *alpha_to,
                                                   const unsigned char *index_of, RsWork *W, int lane)
{
    const int q = lane & 31, base = lane & 32;
    // ---- locator
    int lam = 0;                    // Lambda_{q+1}, polynomial form
    int bq = (q == 0) ? 0 : A0;     // B_q, index form; B(x) = 1 to start with
    int el = 0;
    for (int r = 1; r <= NROOTS; r++) {
        int term = 0;
        if (q <= r - 2 && lam != 0) {
            const int sx = syn[r - 2 - q];
            if (sx != A0) term = alpha_to[gf_mod255((int)index_of[lam] + sx)];
        }
        const int s0 = syn[r - 1];
        const int delta = xor_reduce32(term) ^ (s0 != A0 ? (int)alpha_to[s0] : 0);
        const int up_b = __shfl(bq, base + ((q + 31) & 31), 64);    // B_{q-1}
        const int up_l = __shfl(lam, base + ((q + 31) & 31), 64);   // Lambda_q
        const int shifted = (q == 0) ? A0 : up_b;                   // coefficient q of x.B(x)
        if (delta == 0) {
            bq = shifted;
        } else {
            const int dx = index_of[delta];
            const int next = lam ^ (bq != A0 ? (int)alpha_to[gf_mod255(dx + bq)] : 0);
            if (2 * el <= r - 1) {
                el = r - el;
                const int lq = (q == 0) ? 1 : up_l;  // coefficient q of the outgoing Lambda
                bq = (lq == 0) ? A0 : gf_mod255((int)index_of[lq] - dx + NN);
            } else {
                bq = shifted;
            }
            lam = next;
        }
    }
    const unsigned nzl = (unsigned)(__ballot(lam != 0) >> base);
    const int deg = nzl ? 32 - __clz(nzl) : 0;
    W->lam[q + 1] = index_of[lam];
    if (q == 0) W->lam[0] = 0;
    FEC_WAVE_SYNC();
    const int deg_other = __shfl(deg, base ^ 32, 64);
    const int deg_max = deg > deg_other ? deg : deg_other;
    // ---- roots: element alpha^i, i = q+1+32k
    int count = 0;
    for (int k = 0; k < 8; k++) {
        const int i = q + 1 + 32 * k;
        int v = 1;
        int e = 0;  // i*j mod 255, stepped
        for (int j = 1; j <= deg_max; j++) {
            e += i;
            if (e >= 255) e -= 255;
            const int lj = W->lam[j];
            if (j <= deg && lj != A0) v ^= alpha_to[gf_mod255(lj + e)];
        }
        const bool hit = (i <= NN) && (v == 0);
        const unsigned m = (unsigned)(__ballot(hit) >> base);
        if (hit) {
            const int slot = count + __popc(m & ((1u << q) - 1u));
            if (slot < NROOTS) {
                W->root[slot] = (short)i;
                W->loc[slot] = (short)gf_mod255(IPRIM * i - 1);
            }
        }
        count += __popc(m);
    }
    FEC_WAVE_SYNC();
    const bool miss = count != deg;  // uniform per half-wave
    // ---- evaluator: Omega_q = sum_j S_{q-j} Lambda_j, j <= min(deg, q)
    int om = 0;
    {
        const int top = deg < q ? deg : q;
        for (int j = 0; j <= deg_max && j <= 31; j++) {
            if (j <= top) {
                const int sx = syn[q - j], lj = W->lam[j];
                if (sx != A0 && lj != A0) om ^= alpha_to[gf_mod255(sx + lj)];
            }
        }
    }
    W->omg[q] = index_of[om];
    const unsigned nzo = (unsigned)(__ballot(om != 0) >> base);
    const int dego = nzo ? 31 - __clz(nzo) : 0;
    FEC_WAVE_SYNC();
    // ---- error values, one root per lane
    bool undefined = false;
    int patch = 0, where = 0;
    const bool mine = !miss && q < count;
    {
        const int rt = mine ? (int)W->root[q] : 0;
        const int dego_other = __shfl(dego, base ^ 32, 64);
        const int n1 = dego > dego_other ? dego : dego_other;
        int num = 0, e = 0;  // e = i*rt mod 255
        for (int i = 0; i <= n1; i++) {
            const int oi = W->omg[i];
            if (i <= dego && oi != A0) num ^= alpha_to[gf_mod255(oi + e)];
            e += rt;
            if (e >= 255) e -= 255;
        }
        int den = 0;
        const int dtop = (deg < NROOTS - 1 ? deg : NROOTS - 1) & ~1;
        const int dtop_max = (deg_max < NROOTS - 1 ? deg_max : NROOTS - 1) & ~1;
        const int r2 = gf_mod255(2 * rt);
        e = 0;  // i*rt mod 255 for even i
        for (int i = 0; i <= dtop_max; i += 2) {
            const int l1 = W->lam[i + 1];
            if (i <= dtop && l1 != A0) den ^= alpha_to[gf_mod255(l1 + e)];
            e += r2;
            if (e >= 255) e -= 255;
        }
        if (mine) {
            undefined = den == 0;
            if (!undefined && num != 0) {
                const int scale = alpha_to[gf_mod255(rt * (FCR - 1) + NN)];
                patch = alpha_to[gf_mod255((int)index_of[num] + (int)index_of[scale] + NN - (int)index_of[den])];
                where = W->loc[q];
            }
        }
    }
    const bool fail = miss || ((unsigned)(__ballot(undefined) >> base) != 0u);
    if (!fail && patch != 0) cw[where] ^= (unsigned char)patch;
    return fail ? -1 : count;
}

// Position of the decision word of trellis step k in the GLOBAL scratch (k_fec_bpsk): the parallel chain-back has
// lane j read the words of bit i = 40 j + c at the same moment, so the words are stored c-major (row c = i % 40, column
// i / 40): one coalesced 512-byte load per step instead of 64 scattered ones.  The forward pass pays with a scattered
// store per 64 steps (nobody waits for stores).  Steps k < 6 carry no output bit and park behind the table.
__device__ __forceinline__ int dec_gpos(int k)
{
    const int i = k - 6;
    return i < 0 ? 40 * 64 + k : (i % 40) * 64 + i / 40;
}

// FECDecode (:703-852) on L.raw; payload to L.data only on success (as the reference leaves
// RSdecdata untouched on failure).  Returns -1 or the channel error count (wave-uniform).
template <int DECW, bool VIT_DONE = false>
__device__ __forceinline__ int fec_decode_wave(FecLdsT<DECW> &L, int lane, unsigned long long *decg = nullptr)
{
    constexpr bool DEC_GLOBAL = DECW != DECW_LDS;  // decision words in the global scratch decg[DEC_SCRATCH_WORDS]
    // (VIT_DONE: k_vit64 has run the Viterbi decoder; L.vit holds its output bytes)
    // ---- steps 1+2: de-interleave (:715-722) fused with the branch metrics (:220-225), 512 trellis steps at
    // a time, then add-compare-select with lane = state (:229-253).  Metrics fit int32 (|m| < 3e6).
    if constexpr (!VIT_DONE) {
        // Syms[i] = (Partab[i&0x4f]<<1) | (1-Partab[i&0x6d])   (:105-114)
        const int ia = (parity7(lane & 0x4f) << 1) | (1 - parity7(lane & 0x6d));
        const int ib = (parity7((lane ^ 1) & 0x4f) << 1) | (1 - parity7((lane ^ 1) & 0x6d));
        int metric = (lane == 0) ? 0 : -999999;
        const int src_lo = lane >> 1, src_hi = (lane >> 1) + 32;
        unsigned dlo = 0, dhi = 0;  // DEC_GLOBAL: lane (k & 63) collects the word of step k; flushed every 64 steps
        constexpr int METS_CHUNK = FecLdsT<DECW>::METS;
        for (int k0 = 0; k0 < NBITS; k0 += METS_CHUNK) {
            const int kn = (NBITS - k0) < METS_CHUNK ? (NBITS - k0) : METS_CHUNK;
            for (int kk = lane; kk < kn; kk += 64) {
                const int j0 = 2 * (k0 + kk), j1 = j0 + 1;
                const int y0 = L.sym((j0 % COLUMNS) * ROWS + (j0 / COLUMNS + 1));
                const int y1 = L.sym((j1 % COLUMNS) * ROWS + (j1 / COLUMNS + 1));
#pragma unroll
                for (int i = 0; i < 4; i++)
                    L.mets[kk][i] = (short)(c_fec.mettab[(i >> 1) & 1][y0] + c_fec.mettab[i & 1][y1]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int kk = 0; kk < kn; kk++) {
                const short *m = L.mets[kk];
                int lo = __shfl(metric, src_lo, 64);
                int hi = __shfl(metric, src_hi, 64);
                int m0 = lo + (int)m[ia];
                int m1 = hi + (int)m[ib];
                bool d = m1 > m0;
                metric = d ? m1 : m0;
                unsigned long long mask = __ballot(d);
                if constexpr (DEC_GLOBAL) {
                    const int k = k0 + kk, slot = k & 63;
                    const bool mine = lane == slot;
                    dlo = mine ? (unsigned)mask : dlo;
                    dhi = mine ? (unsigned)(mask >> 32) : dhi;
                    if ((slot == 63 || k == NBITS - 1) && (k & ~63) + lane < NBITS)  // one store per 64 trellis steps
                        decg[dec_gpos((k & ~63) + lane)] = ((unsigned long long)dhi << 32) | dlo;
                } else {
                    if (lane == 0) L.dec[k0 + kk] = mask;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // ---- step 2c: chain back from state 0 (:264-276); decision words fetched 64 steps at a time
    if constexpr (!VIT_DONE) {
        for (int i = lane; i < 320; i += 64) L.vit[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int beststate = 0;
        int cur = 0;  // byte being assembled (bits of 8 consecutive i)
        // DEC_GLOBAL: the words were written by this wave; they are read back past the L1 (a grid-stride
        // caller may have cached the same scratch lines from an earlier block), one chunk ahead of their use
        auto fetch = [&](int hi) -> unsigned long long {
            const int my = hi - lane;
            if (hi < 0 || my < 0) return 0ull;
            if constexpr (DEC_GLOBAL)
                return __hip_atomic_load(&decg[dec_gpos(my + 6)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                return L.dec[my + 6];
        };
        if constexpr (DEC_GLOBAL) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the decision stores have reached L2
            __builtin_amdgcn_wave_barrier();
        }
        // The chain-back is one dependent step per bit (2560 links of ~140 cycles).  Survivor paths merge within a few
        // constraint lengths, so every lane traces ITS OWN 40 bits (5 bytes) after a warm-up of WARM steps from an
        // arbitrary state above its segment, all 64 lanes in lock-step: 104 steps instead of 2560.  The result is
        // accepted only if the segments chain up exactly -- lane j enters its segment in the state lane j+1 left its own
        // (lane 63 starts in the true state 0), which by induction makes every lane's path the reference's path;
        // otherwise (paths that did not merge: garbage frames) the serial chain-back below runs as before.
        bool par_ok = false;
        {
            constexpr int SEG = 40, WARM = 128, LAST = NBITS - 7;  // bits 0..LAST (warm-up 128: blocks with ~8 % symbol errors still merge)
            static_assert(64 * SEG == LAST + 1, "64 lanes x 40 bits");
            const int seg_lo = SEG * lane;
            int st = 0, top = 0;
            unsigned long long bits = 0;  // bit (i - seg_lo) of the segment
            for (int t0 = 0; t0 < SEG + WARM; t0 += 8) {
                unsigned long long w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {  // the eight words in flight together: they do not depend on the state
                    const int i = seg_lo + SEG - 1 + WARM - (t0 + u);
                    const int ic = i > LAST ? LAST : i;
                    if constexpr (DEC_GLOBAL)
                        w[u] = __hip_atomic_load(&decg[dec_gpos(ic + 6)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        w[u] = L.dec[ic + 6];
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int i = seg_lo + SEG - 1 + WARM - (t0 + u);
                    if (i == seg_lo + SEG - 1) top = st;
                    if (i <= LAST) {
                        const unsigned word = (st >> 5) ? (unsigned)(w[u] >> 32) : (unsigned)w[u];
                        const unsigned bit = (word >> (st & 31)) & 1u;
                        st = (int)(((unsigned)st | (bit << 6)) >> 1);
                        if (i <= seg_lo + SEG - 1) bits |= (unsigned long long)bit << (i - seg_lo);
                    }
                }
            }
            const int above = __shfl_down(st, 1, 64);  // the state in which lane j+1 left its segment
            const bool good = (lane == 63) ? (top == 0) : (top == above);
            par_ok = __ballot(!good) == 0ull;
            if (par_ok) {
#pragma unroll
                for (int b = 0; b < SEG / 8; b++)  // bit i sits at 0x80 >> (i & 7) of byte i >> 3 (:270-273)
                    L.vit[(seg_lo >> 3) + b] = (unsigned char)(__brev((unsigned)((bits >> (8 * b)) & 0xffu)) >> 24);
            }
        }
        unsigned long long wnext = par_ok ? 0ull : fetch(NBITS - 7);
        for (int hi_i = par_ok ? -1 : NBITS - 7; hi_i >= 0; hi_i -= 64) {
            // this chunk covers i = hi_i .. max(hi_i-63,0); step index k = i + 6
            unsigned long long w = wnext;
            wnext = fetch(hi_i - 64);
            unsigned wl = (unsigned)w, wh = (unsigned)(w >> 32);
            int nsteps = hi_i + 1 < 64 ? hi_i + 1 : 64;
            for (int t = 0; t < nsteps; t++) {
                int i = hi_i - t;
                unsigned l32 = __builtin_amdgcn_readlane(wl, t);
                unsigned h32 = __builtin_amdgcn_readlane(wh, t);
                unsigned word = (beststate >> 5) ? h32 : l32;
                if ((word >> (beststate & 31)) & 1u) {
                    beststate |= 64;
                    cur |= 0x80 >> (i & 7);
                }
                beststate >>= 1;
                if ((i & 7) == 0) {
                    if (lane == 0) L.vit[i >> 3] = (unsigned char)cur;
                    cur = 0;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- step 3: de-scramble into the two RS code words (:765-771), syndromes (:336-347)
    const int blk = lane >> 5, ri = lane & 31;
    for (int i = lane; i < 2 * 256; i += 64) (&L.rs[0][0])[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 320; i += 64) L.rs[i & 1][RSPAD + (i >> 1)] = L.vit[i] ^ c_fec.scrambler[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int syn = 0;
    {
        const int step = (FCR + ri) * PRIM;
        const unsigned char *dw = L.rs[blk];
        for (int j = RSPAD; j < NN; j++) {  // columns < RSPAD are zero padding: s stays 0 through them
            int dj = dw[j];
            syn = (syn == 0) ? dj : (dj ^ (int)L.alpha_to[gf_mod255((int)L.index_of[syn] + step)]);
        }
    }
    unsigned long long nz = __ballot(syn != 0);
    int *sidx = reinterpret_cast<int *>(&L.mets[0][0]);  // branch metrics are dead: reuse as int s[2][32]
    sidx[lane] = L.index_of[syn];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int rserr = 0;
    if (nz != 0ull) {  // a code word with all-zero syndromes is left alone and counts 0 corrections (:351-358)
        const bool need = ((nz >> (lane & 32)) & 0xffffffffull) != 0ull;
        const int r = rs_correct_halfwave(L.rs[blk], sidx + blk * 32, L.alpha_to, L.index_of,
                                          reinterpret_cast<RsWork *>(reinterpret_cast<short *>(&L.dec[0]) + blk * 272), lane);
        rserr = need ? r : 0;  // decisions are dead by now: their LDS holds the work areas
    }
    int e0 = __shfl(rserr, 0, 64), e1 = __shfl(rserr, 32, 64);
    FEC_WAVE_SYNC();
    if (e0 == -1 || e1 == -1) return -1;  // :821-824
    for (int j = lane; j < 256; j += 64) L.data[j] = L.rs[j & 1][RSPAD + (j >> 1)];  // :783-787
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- step 4: re-encode and count channel errors (:831-847)
    fec_encode_wave(L, L.data, lane);
    int errs = 0;
    {
        const unsigned char *enc = fec_enc(L);
        for (int i = lane; i < SYMPBLOCK; i += 64) errs += (enc[i] != (L.sym(i) >> 7)) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) errs += __shfl_xor(errs, off, 64);
    return errs;
}

template <int DECW>
__device__ __forceinline__ void fec_lds_init(FecLdsT<DECW> &L, int lane)
{
    for (int i = lane; i < 256; i += 64) {
        L.alpha_to[i] = c_fec.alpha_to[i];
        L.index_of[i] = c_fec.index_of[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void k_fec_decode(const unsigned char *__restrict__ raw, long long nblocks,
                                                   unsigned char *__restrict__ out, int *__restrict__ rc)
{
    __shared__ FecLds L;
    const int lane = threadIdx.x;
    fec_lds_init(L, lane);
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const unsigned *src = reinterpret_cast<const unsigned *>(raw + blk * SYMPBLOCK);
        for (int i = lane; i < SYMPBLOCK / 4; i += 64) reinterpret_cast<unsigned *>(L.raw)[i] = src[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int r = fec_decode_wave(L, lane);
        if (r >= 0)
            for (int i = lane; i < 256; i += 64) out[blk * 256 + i] = L.data[i];
        if (lane == 0) rc[blk] = r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(64) void k_fec_encode(const unsigned char *__restrict__ data, long long nblocks,
                                                   unsigned char *__restrict__ sym)
{
    __shared__ FecLds L;
    const int lane = threadIdx.x;
    fec_lds_init(L, lane);
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        for (int i = lane; i < 256; i += 64) L.data[i] = data[blk * 256 + i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        fec_encode_wave(L, L.data, lane);
        unsigned *dst = reinterpret_cast<unsigned *>(sym + blk * SYMPBLOCK);
        for (int i = lane; i < SYMPBLOCK / 4; i += 64) dst[i] = reinterpret_cast<unsigned *>(fec_enc(L))[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// stage 2 of the BPSK hook, per stream, in hit order: a successful decode replaces decoded[] (:565-569); a failed one
// leaves it (FECDecoder.java:780), and its log entry shows the bytes the demodulator still holds.  One wave.
__device__ __forceinline__ void fec_bpsk_fin(const BpskFecArgs &a, int s, int nt, int lane)
{
    unsigned char *dst = a.decoded + (long long)s * 256;
    unsigned cur = reinterpret_cast<unsigned *>(dst)[lane];  // 256 bytes = one dword per lane
    int ndec = 0, lastrc = 0;
    for (int t = 0; t < nt; t++) {
        unsigned *logd = reinterpret_cast<unsigned *>(a.fec_data + ((long long)s * a.max_trig + t) * 256);
        const int r = a.fec_rc[s * a.max_trig + t];
        if (r >= 0) {
            cur = logd[lane];
            ndec++;
        } else {
            logd[lane] = cur;
        }
        lastrc = r;
    }
    reinterpret_cast<unsigned *>(dst)[lane] = cur;
    if (lane == 0) {
        a.last[2 * s] = lastrc;
        a.last[2 * s + 1] = lastrc < 0 ? 0 : 1;
        a.cnt_dec[s] += ndec;
    }
}

__device__ __forceinline__ void fec_post_copies(const BpskFecArgs &a, int lane)
{
    for (int c = 0; c < a.ncopy; c++)
        for (int i = lane; i < a.cbytes[c]; i += 64) a.cdst[c][i] = a.csrc[c][i];
}

// BPSK hook: one wave per (stream, sync hit).  Build the hard-decision block from the +1/-1/0 bit history
// (FUNcubeBPSKDemod.java:562-564) and decode; payload (on success) and rc go to the per-hit log.  The block that
// finishes LAST for its stream (a device-scope counter per stream; release / acquire fences around it) then runs stage 2
// for that stream when a.fuse is set (short single-stream calls: one dependent launch less); otherwise k_fec_fin does.
__global__ __launch_bounds__(64) void k_fec_bpsk(BpskFecArgs a)
{
    __shared__ FecLdsT<DECW_WORK> L;
    const int lane = threadIdx.x;
    // x = stream, y = hit index: consecutive workgroups go to different XCDs, and the few hits per stream
    // (y small) must not all land on the same one or two XCDs
    const int s = blockIdx.x, t = blockIdx.y;
    int nt = a.trig_count[s];
    if (nt > a.max_trig) nt = a.max_trig;
    if (t >= nt) {
        if (a.fuse && nt == 0 && t == 0) fec_post_copies(a, lane);  // no hit in this call: nothing to wait for
        return;
    }
    fec_lds_init(L, lane);
    const signed char *win = a.bitlog + (long long)s * a.bitlog_stride + (a.trig_bits[s * a.max_trig + t] + 1);
    // hard decisions, one bit per symbol: 64 symbols per ballot, lane 0 files the word
    for (int i0 = 0; i0 < SYMPBLOCK; i0 += 64) {
        const int i = i0 + lane;
        const unsigned long long m = __ballot(i < SYMPBLOCK && win[i < SYMPBLOCK ? i : SYMPBLOCK - 1] == 1);
        if (lane == 0) reinterpret_cast<unsigned long long *>(L.raw)[i0 >> 6] = m;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = fec_decode_wave(L, lane, a.dec_scratch + ((long long)s * a.max_trig + t) * DEC_SCRATCH_WORDS);
    unsigned char *logd = a.fec_data + ((long long)s * a.max_trig + t) * 256;
    if (r >= 0)
        for (int i = lane; i < 256; i += 64) logd[i] = L.data[i];
    if (lane == 0) a.fec_rc[s * a.max_trig + t] = r;
    if (!a.fuse) return;  // (uniform) k_fec_fin follows
    // ---- the stream's last block to get here runs stage 2
    __threadfence();  // release: this block's log entry is visible device-wide before the count goes up
    int prev = 0;
    if (lane == 0) prev = atomicAdd(a.done + s, 1);
    prev = __shfl(prev, 0, 64);
    if (prev == nt - 1) {
        __threadfence();  // acquire: the other blocks' log entries
        fec_bpsk_fin(a, s, nt, lane);
        if (lane == 0) a.done[s] = 0;  // ready for the next launch
        FEC_WAVE_SYNC();  // stage 2's stores (other lanes') before the copies read them
        fec_post_copies(a, lane);
    }
}

// ---------------------------------------------------------------------------------------------- k_vitq
// The batch form of the demodulator's hook: FOUR LANES PER BLOCK for the Viterbi decoder, sixteen blocks per wave.
// With lane = state (above) a trellis step costs ~20 wave instructions for ONE block -- two cross-lane reads through the
// LDS crossbar, the ballot, the bookkeeping of the decision word -- and a lone wave waits out every one of their latencies.
// Here lane d of a quad keeps the 16 path metrics of the states n = 4m + d (m = register index) of ITS block.  The
// predecessors of state 4m + d are 2m + (d >> 1) and that + 32: registers m >> 1 and (m >> 1) + 8 of quad lane
// 2 (m & 1) + (d >> 1) -- a FIXED quad permutation per register ([0,0,1,1] for even m, [2,2,3,3] for odd m), which the
// DPP operand of the add performs on the way: no instruction is spent on moving data.  The branch symbols
// Syms[n] (:105-114) are linear in the bits of n, Syms[4m + d] = Syms[4m] ^ G(d), so a lane keeps its own permutation of
// the step's four branch metrics and indexes it with the compile-time Syms[4m].  A step = 4 selects + 4 adds for the
// branch metrics and, per register, 2 adds, 1 compare whose result an add-with-carry shifts into the lane's decision
// word, 1 maximum: ~100 wave instructions for 16 blocks = 6 per block instead of 20, and sixteen times fewer waves
// than blocks.  Same integer metrics (mettab sums, :220-225), same strict comparison m1 > m0 (:240-251), same decisions;
// the chain-back (:264-276) runs in the same wave, the four lanes of a quad following the same path.
// (Tried first: ONE lane per block, all 64 metrics in its registers, 64 blocks per wave -- no cross-lane traffic at all, 6
// instructions per block and step as well, but 261 waves for the 16 700 blocks of an 8192-stream call: 3.5 ms of latency
// on a quarter of the SIMDs against 1.97 ms for the lane = state kernel, and beside the main stream's kernels its 159
// VGPRs + 41 KB of LDS cost every CU it sat on one of their workgroups.)
//   k_fec_list : the (stream, hit) pairs of the call, compacted into one work list
//   k_vitq     : symbols -> hard bits in trellis order (whole wave per block, ballots; LDS), ACS, chain-back -> the 320
//                output bytes per block; decision words through a global scratch [4 steps][lane] (coalesced both ways)
//   k_fec_rs   : one wave per block, as k_fec_bpsk from its step 3: RS syndromes / correction, re-encode, error count
struct VitArgs {
    const signed char *bitlog;
    long long bitlog_stride;
    const int *trig_count, *trig_bits;
    int max_trig, nstreams;
    int *work_list;            // [nstreams * max_trig] item = stream * max_trig + hit
    int *work_count;           // [1]
    unsigned long long *dec;   // [waves][VQ_GROUPS][64]
    unsigned char *vit;        // [nstreams * max_trig][320]
    int quad_cap;              // blocks one round of quad waves takes: 16 x the device's SIMDs
    unsigned long long *rem_dec;  // [VQ_REM_MAX][DEC_SCRATCH_WORDS] decision words of the remainder's blocks
};
// How many of the nwork blocks take the quad kernel.  A quad wave issues ~100 instructions per trellis step whatever the
// number of its blocks, and it fills its SIMD's issue slots: 1044 of them on 1024 SIMDs take as long as 2048.  So the blocks
// beyond a whole number of rounds -- when they are few -- go through the lane = state decoder (one wave per block: latency
// bound, its waves fit between the quad waves' instructions) in the SAME launch.
enum { VQ_REM_MAX = 4096 };
__host__ __device__ inline int vq_quad_blocks(int nwork, int quad_cap)
{
    const int rem = nwork % quad_cap;
    return rem < VQ_REM_MAX && nwork >= quad_cap ? nwork - rem : nwork;
}
enum { VQ_BLOCKS = 16, VQ_GROUPS = (NBITS + 3) / 4 };

// A block's 5200 symbols as hard bits (FUNcubeBPSKDemod.java:562-564: bit == 1 -> 0xc0, else 0x40), one wave per block:
//   words 0 .. 80   : in the order the trellis takes them (de-interleaver, :715-722: symbols 2k, 2k+1 of step k sit at
//                     (j % 65) * 80 + j / 65 + 1), for k_vitq
//   words 81 .. 162 : in their own order, for k_fec_rs (what k_fec_bpsk keeps in L.raw)
// The window comes in with all its loads in flight (dwords around the byte-aligned start) and is looked up in LDS: as 163
// dependent global byte gathers per wave this took longer than the trellis.
enum { VQ_SW = 81, VQ_NW = (SYMPBLOCK + 63) / 64, VQ_BITS = VQ_SW + VQ_NW };
__global__ __launch_bounds__(64) void k_fec_bits(VitArgs a, unsigned long long *bits)
{
    __shared__ unsigned winL[SYMPBLOCK / 4 + 4];
    const int lane = threadIdx.x;
    const int nwork = __builtin_amdgcn_readfirstlane(*a.work_count);
    const int idx = blockIdx.x;
    if (idx >= nwork) return;
    const int item = __builtin_amdgcn_readfirstlane(a.work_list[idx]);
    const int s = item / a.max_trig;
    const signed char *win = a.bitlog + (long long)s * a.bitlog_stride + (a.trig_bits[item] + 1);
    const int sh = (int)(reinterpret_cast<unsigned long long>(win) & 3ull);
    const unsigned *w32 = reinterpret_cast<const unsigned *>(win - sh);
    constexpr int NDW = SYMPBLOCK / 4 + 1, NIT = (NDW + 63) / 64;  // (+1: the start may sit inside a dword)
    unsigned v[NIT];
#pragma unroll
    for (int q = 0; q < NIT; q++) {
        const int i = lane + 64 * q;
        v[q] = w32[i < NDW ? i : NDW - 1];
    }
#pragma unroll
    for (int q = 0; q < NIT; q++) {
        const int i = lane + 64 * q;
        if (i < NDW) winL[i] = v[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const signed char *wl = reinterpret_cast<const signed char *>(winL) + sh;  // wl[i] == win[i]
    unsigned long long *dst = bits + (long long)item * VQ_BITS;
    unsigned long long m0 = 0ull, m1 = 0ull, m2 = 0ull;  // lane q keeps word q, q + 64, q + 128
#pragma unroll 9
    for (int q = 0; q < VQ_SW; q++) {
        const int j = 64 * q + lane;
        const int jc = j < 2 * NBITS ? j : 2 * NBITS - 1;
        const int src = (jc % COLUMNS) * ROWS + (jc / COLUMNS + 1);
        const unsigned long long m = __ballot(j < 2 * NBITS && wl[src] == 1);
        if (q < 64) m0 = (lane == q) ? m : m0; else m1 = (lane == q - 64) ? m : m1;
    }
#pragma unroll 2
    for (int q = 0; q < VQ_NW; q++) {
        const int i = 64 * q + lane;
        const unsigned long long m = __ballot(i < SYMPBLOCK && wl[i < SYMPBLOCK ? i : SYMPBLOCK - 1] == 1);
        const int w = VQ_SW + q;
        if (w < 128) m1 = (lane == w - 64) ? m : m1; else m2 = (lane == w - 128) ? m : m2;
    }
    dst[lane] = m0;
    dst[64 + lane] = m1;
    if (128 + lane < VQ_BITS) dst[128 + lane] = m2;
}

__global__ void k_fec_list(VitArgs a)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.nstreams) return;
    int nt = a.trig_count[s];
    if (nt > a.max_trig) nt = a.max_trig;
    if (nt <= 0) return;
    const int base = atomicAdd(a.work_count, nt);
    for (int t = 0; t < nt; t++) a.work_list[base + t] = s * a.max_trig + t;
}

__host__ __device__ constexpr int vit_par7(int v)
{
    int p = 0;
    for (int b = 0; b < 7; b++) p ^= (v >> b) & 1;
    return p;
}
// Syms[n] = (Partab[n&0x4f]<<1) | (1-Partab[n&0x6d])   (:105-114): the symbol pair on the branch INTO state n from n>>1
__host__ __device__ constexpr int vit_ia(int n) { return (vit_par7(n & 0x4f) << 1) | (1 - vit_par7(n & 0x6d)); }
static_assert((vit_ia(4 * 5 + 1) == (vit_ia(4 * 5) ^ 3)) && (vit_ia(4 * 9 + 2) == (vit_ia(4 * 9) ^ 2)) && (vit_ia(4 * 14 + 3) == (vit_ia(4 * 14) ^ 1)),
              "Syms[4m + d] = Syms[4m] ^ G(d), G = 0, 3, 2, 1");

template <int CTRL>
__device__ __forceinline__ int vq_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// one trellis step of the quad's block: O -> N; returns the lane's 16 decisions (bit m: state 4m + d)
__device__ __forceinline__ unsigned vitq_step(const int (&O)[16], int (&N)[16], const int (&pb)[4])
{
    unsigned dw = 0u;
#pragma unroll
    for (int m = 15; m >= 0; m--) {  // shifted in from the top register down: bit m ends at position m
        constexpr int EVEN = 0x50, ODD = 0xfa;  // quad_perm [0,0,1,1] / [2,2,3,3]
        const int x0 = (m & 1) ? vq_dpp<ODD>(O[m >> 1]) : vq_dpp<EVEN>(O[m >> 1]);
        const int x1 = (m & 1) ? vq_dpp<ODD>(O[(m >> 1) + 8]) : vq_dpp<EVEN>(O[(m >> 1) + 8]);
        const int m0 = x0 + pb[vit_ia(4 * m)];
        const int m1 = x1 + pb[vit_ia(4 * m) ^ 3];
        // dw = 2 dw + (m1 > m0): the compare's result goes in as the carry of dw + dw (the compiler's own form is a select
        // and a shift-or per bit, with wait states behind every compare: 135 instead of 100 instructions a step)
        asm("v_cmp_gt_i32_e32 vcc, %2, %3\n\tv_addc_co_u32_e32 %0, vcc, %1, %1, vcc" : "=v"(dw) : "v"(dw), "v"(m1), "v"(m0) : "vcc");
        N[m] = m1 > m0 ? m1 : m0;
    }
    return dw;
}

__global__ __launch_bounds__(64) void k_vitq(VitArgs a, const unsigned long long *bits, BpskFecArgs fa)
{
    const int lane = threadIdx.x, quad = lane >> 2, d = lane & 3;
    const int nwork_all = __builtin_amdgcn_readfirstlane(*a.work_count);
    const int nwork = vq_quad_blocks(nwork_all, a.quad_cap);  // the list's first nwork blocks are the quad waves'
    const int base = blockIdx.x * VQ_BLOCKS;
    if (base >= nwork) {
        // ---- the remainder: one wave per block, the whole of FECDecode (what k_fec_bpsk does)
        const int ridx = (int)blockIdx.x - (nwork + VQ_BLOCKS - 1) / VQ_BLOCKS;
        if (nwork + ridx >= nwork_all) return;
        __shared__ FecLdsT<DECW_WORK> L;
        const int item = __builtin_amdgcn_readfirstlane(a.work_list[nwork + ridx]);
        const int s = item / a.max_trig, t = item % a.max_trig;
        fec_lds_init(L, lane);
        {
            const unsigned long long *nat = bits + (long long)item * VQ_BITS + VQ_SW;
            unsigned long long *raw = reinterpret_cast<unsigned long long *>(L.raw);
            raw[lane] = nat[lane];
            if (64 + lane < VQ_NW) raw[64 + lane] = nat[64 + lane];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int r = fec_decode_wave(L, lane, a.rem_dec + (long long)ridx * DEC_SCRATCH_WORDS);
        unsigned char *logd = fa.fec_data + ((long long)s * a.max_trig + t) * 256;
        if (r >= 0)
            for (int i = lane; i < 256; i += 64) logd[i] = L.data[i];
        if (lane == 0) fa.fec_rc[s * a.max_trig + t] = r;
        return;
    }
    const int nit = nwork - base < VQ_BLOCKS ? nwork - base : VQ_BLOCKS;
    const int my = quad < nit ? quad : nit - 1;  // (surplus quads shadow the last block and store nothing)
    const bool live = quad < nit;
    // ---- add-compare-select (:229-253).  The lane's permutation of the branch metrics: pb[k] = bm[k ^ G(d)],
    // bm[i] = mettab[i >> 1][y0] + mettab[i & 1][y1] (:220-225)
    const int A0 = c_fec.mettab[0][0x40], A1 = c_fec.mettab[0][0xc0];  // a received 0 / 1 against a sent 0
    const int B0 = c_fec.mettab[1][0x40], B1 = c_fec.mettab[1][0xc0];  // ... against a sent 1
    const bool g1 = d == 1 || d == 2, g0 = d == 1 || d == 3;           // G(d) = 0, 3, 2, 1
    const int T00 = g1 ? B0 : A0, T01 = g1 ? B1 : A1, T10 = g1 ? A0 : B0, T11 = g1 ? A1 : B1;
    const int U00 = g0 ? B0 : A0, U01 = g0 ? B1 : A1, U10 = g0 ? A0 : B0, U11 = g0 ? A1 : B1;
    int M[16], N[16];
#pragma unroll
    for (int m = 0; m < 16; m++) M[m] = (m == 0 && d == 0) ? 0 : -999999;
    const int item = a.work_list[base + my];
    const unsigned *symw = reinterpret_cast<const unsigned *>(bits + (long long)item * VQ_BITS);  // (k_fec_bits)
    unsigned long long *decw = a.dec + (long long)blockIdx.x * (VQ_GROUPS * 64) + lane;
    auto two_steps = [&](unsigned w, unsigned &d0, unsigned &d1) {  // M -> N -> M, symbol bits 0..3 of w
        int pb[4];
        {
            const int t0 = (w & 1u) ? T01 : T00, t1 = (w & 1u) ? T11 : T10;
            const int u0 = (w & 2u) ? U01 : U00, u1 = (w & 2u) ? U11 : U10;
            pb[0] = t0 + u0;
            pb[1] = t0 + u1;
            pb[2] = t1 + u0;
            pb[3] = t1 + u1;
        }
        d0 = vitq_step(M, N, pb);
        {
            const int t0 = (w & 4u) ? T01 : T00, t1 = (w & 4u) ? T11 : T10;
            const int u0 = (w & 8u) ? U01 : U00, u1 = (w & 8u) ? U11 : U10;
            pb[0] = t0 + u0;
            pb[1] = t0 + u1;
            pb[2] = t1 + u0;
            pb[3] = t1 + u1;
        }
        d1 = vitq_step(N, M, pb);
    };
    constexpr int NQ = (2 * NBITS + 31) / 32;  // 161 dwords of symbol bits: 160 x 16 steps + 6
    static_assert(NBITS - 16 * (NQ - 1) == 6 && VQ_GROUPS == 4 * (NQ - 1) + 2, "the last dword holds six steps: a group of four, a group of two");
    unsigned wn = symw[0];
#pragma unroll 1
    for (int q = 0; q < NQ; q++) {
        unsigned w = wn;
        wn = symw[q + 1 < NQ ? q + 1 : q];  // (one dword = sixteen steps ahead)
        const int ng = q < NQ - 1 ? 4 : 2;
#pragma unroll 1
        for (int g = 0; g < ng; g++) {
            unsigned d0, d1, d2 = 0u, d3 = 0u;
            two_steps(w, d0, d1);
            if (q < NQ - 1 || g == 0) two_steps(w >> 4, d2, d3);  // (the call's last group holds two steps)
            if (live) decw[(long long)(4 * q + g) * 64] = ((unsigned long long)(d2 | (d3 << 16)) << 32) | (d0 | (d1 << 16));
            w >>= 8;
        }
    }
    // ---- chain back from state 0 (:264-276).  Bit i of the output is decided at trellis step k = i + 6: group k >> 2,
    // field k & 3 of the word of quad lane (state & 3), bit state >> 2.  The four lanes follow the same path: each looks the
    // bit up in ITS word, the quad ORs the four candidates together (two DPP steps) and picks lane state & 3's.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the words were written by this wave; read back past the L1
    __builtin_amdgcn_wave_barrier();
    unsigned char *vitp = a.vit + (long long)item * 320;
    int st = 0;
    unsigned cur = 0;
#pragma unroll 1
    for (int gb = VQ_GROUPS - 1; gb >= 0; gb -= 8) {
        unsigned long long W[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int g = gb - u;
            W[u] = __hip_atomic_load(&decw[(long long)(g < 0 ? 0 : g) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int t = 3; t >= 0; t--) {
                const int k = 4 * (gb - u) + t, i = k - 6;  // (uniform)
                if (i >= 0 && i <= NBITS - 7) {
                    const unsigned f = (unsigned)(W[u] >> (16 * t)) & 0xffffu;
                    unsigned c4 = ((f >> (st >> 2)) & 1u) << d;
                    c4 |= (unsigned)vq_dpp<0xb1>((int)c4);  // quad_perm [1,0,3,2]
                    c4 |= (unsigned)vq_dpp<0x4e>((int)c4);  // quad_perm [2,3,0,1]
                    const unsigned bit = (c4 >> (st & 3)) & 1u;
                    st = (int)(((unsigned)st | (bit << 6)) >> 1);
                    cur |= bit << (7 - (i & 7));  // bit i sits at 0x80 >> (i & 7)  (:270-273)
                    if ((i & 7) == 0) {
                        if (live && d == 0) vitp[i >> 3] = (unsigned char)cur;
                        cur = 0;
                    }
                }
            }
        }
    }
}

// stage 1b of the batch form: everything of FECDecode behind the Viterbi decoder, one wave per (stream, hit)
__global__ __launch_bounds__(64) void k_fec_rs(BpskFecArgs a, const unsigned char *vit, const unsigned long long *bits, int quad_cap)
{
    __shared__ FecLdsT<DECW_WORK> L;
    const int lane = threadIdx.x;
    const int nwork = vq_quad_blocks(__builtin_amdgcn_readfirstlane(*a.work_count), quad_cap);  // (the others are done)
    const int idx = blockIdx.x;
    if (idx >= nwork) return;
    const int item = __builtin_amdgcn_readfirstlane(a.work_list[idx]);
    const int s = item / a.max_trig, t = item % a.max_trig;
    fec_lds_init(L, lane);
    {
        const unsigned long long *nat = bits + (long long)item * VQ_BITS + VQ_SW;  // the symbols' hard bits in their own order
        unsigned long long *raw = reinterpret_cast<unsigned long long *>(L.raw);
        raw[lane] = nat[lane];
        if (64 + lane < VQ_NW) raw[64 + lane] = nat[64 + lane];
    }
    const unsigned char *v = vit + ((long long)s * a.max_trig + t) * 320;
    for (int i = lane; i < 320; i += 64) L.vit[i] = v[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int r = fec_decode_wave<DECW_WORK, true>(L, lane, nullptr);
    unsigned char *logd = a.fec_data + ((long long)s * a.max_trig + t) * 256;
    if (r >= 0)
        for (int i = lane; i < 256; i += 64) logd[i] = L.data[i];
    if (lane == 0) a.fec_rc[s * a.max_trig + t] = r;
}

__global__ __launch_bounds__(256) void k_fec_fin(BpskFecArgs a)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= a.nstreams) return;
    int nt = a.trig_count[s];
    if (nt > a.max_trig) nt = a.max_trig;
    if (nt <= 0) return;
    fec_bpsk_fin(a, s, nt, lane);
}

int fec_dec_scratch_words() { return DEC_SCRATCH_WORDS; }

static long long fec_vitq_dec_words(int nstreams, int max_trig)
{
    const long long waves = ((long long)nstreams * max_trig + VQ_BLOCKS - 1) / VQ_BLOCKS;
    return waves * 64 * VQ_GROUPS;
}
// decision words of every quad wave + the hard bits of every block + the decision words of the remainder's blocks
long long fec_vitq_scratch_words(int nstreams, int max_trig)
{
    return fec_vitq_dec_words(nstreams, max_trig) + (long long)nstreams * max_trig * VQ_BITS + (long long)VQ_REM_MAX * DEC_SCRATCH_WORDS;
}

static int simd_count()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
            (void)hipGetLastError();
            cus = 256;
        }
        n = 4 * cus;
    }
    return n;
}

int launch_fec_bpsk(const BpskFecArgs &a, hipStream_t st)
{
    if (upload_tables() != JSDR_OK) return JSDR_ERR;
    if (a.vit && !a.fuse) {  // the batch form: four lanes per block
        VitArgs v;
        v.bitlog = a.bitlog;
        v.bitlog_stride = a.bitlog_stride;
        v.trig_count = a.trig_count;
        v.trig_bits = a.trig_bits;
        v.max_trig = a.max_trig;
        v.nstreams = a.nstreams;
        v.work_list = a.work_list;
        v.work_count = a.work_count;
        v.dec = a.dec_scratch;
        v.vit = a.vit;
        JSDR_HIP_TRY(hipMemsetAsync(a.work_count, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_fec_list, dim3((unsigned)((a.nstreams + 255) / 256)), dim3(256), 0, st, v);
        JSDR_LAUNCH_CHECK();
        const long long waves = ((long long)a.nstreams * a.max_trig + VQ_BLOCKS - 1) / VQ_BLOCKS;
        unsigned long long *bits = a.dec_scratch + fec_vitq_dec_words(a.nstreams, a.max_trig);  // behind the decision words
        const unsigned nmax = (unsigned)((long long)a.nstreams * a.max_trig);  // (one workgroup per POSSIBLE block; those behind the list's end leave at once)
        hipLaunchKernelGGL(k_fec_bits, dim3(nmax), dim3(64), 0, st, v, bits);
        JSDR_LAUNCH_CHECK();
        v.quad_cap = VQ_BLOCKS * simd_count();
        v.rem_dec = bits + (long long)a.nstreams * a.max_trig * VQ_BITS;
        hipLaunchKernelGGL(k_vitq, dim3((unsigned)(waves + VQ_REM_MAX)), dim3(64), 0, st, v, (const unsigned long long *)bits, a);
        JSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_fec_rs, dim3(nmax), dim3(64), 0, st, a, (const unsigned char *)a.vit, (const unsigned long long *)bits, v.quad_cap);
        JSDR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_fec_fin, dim3((unsigned)((a.nstreams + 3) / 4)), dim3(256), 0, st, a);
        JSDR_LAUNCH_CHECK();
        return JSDR_OK;
    }
    hipLaunchKernelGGL(k_fec_bpsk, dim3((unsigned)a.nstreams, (unsigned)a.max_trig), dim3(64), 0, st, a);
    JSDR_LAUNCH_CHECK();
    if (!a.fuse) {
        hipLaunchKernelGGL(k_fec_fin, dim3((unsigned)((a.nstreams + 3) / 4)), dim3(256), 0, st, a);
        JSDR_LAUNCH_CHECK();
    }
    return JSDR_OK;
}

}  // namespace jsdr

using namespace jsdr;

extern "C" {

int jsdr_fec_decode_batch(const uint8_t *raw_dev, int64_t nblocks, uint8_t *out_dev, int32_t *rc_dev, void *stream)
{
    JSDR_REQUIRE(raw_dev && out_dev && rc_dev, "jsdr_fec_decode_batch: null buffer");
    JSDR_REQUIRE(nblocks >= 0, "jsdr_fec_decode_batch: negative block count");
    if (nblocks == 0) return JSDR_OK;
    if (upload_tables() != JSDR_OK) return JSDR_ERR;
    int grid = (int)(nblocks < 4096 ? nblocks : 4096);
    hipLaunchKernelGGL(k_fec_decode, dim3(grid), dim3(64), 0, as_stream(stream), raw_dev, (long long)nblocks, out_dev,
                       rc_dev);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_fec_encode_batch(const uint8_t *data_dev, int64_t nblocks, uint8_t *sym_dev, void *stream)
{
    JSDR_REQUIRE(data_dev && sym_dev, "jsdr_fec_encode_batch: null buffer");
    JSDR_REQUIRE(nblocks >= 0, "jsdr_fec_encode_batch: negative block count");
    if (nblocks == 0) return JSDR_OK;
    if (upload_tables() != JSDR_OK) return JSDR_ERR;
    int grid = (int)(nblocks < 4096 ? nblocks : 4096);
    hipLaunchKernelGGL(k_fec_encode, dim3(grid), dim3(64), 0, as_stream(stream), data_dev, (long long)nblocks, sym_dev);
    JSDR_LAUNCH_CHECK();
    return JSDR_OK;
}

int jsdr_fec_decode(const uint8_t raw_host[5200], uint8_t out_host[256], int *rc)
{
    JSDR_REQUIRE(raw_host && out_host && rc, "jsdr_fec_decode: null argument");
    DevBuf<unsigned char> raw, out;
    DevBuf<int> drc;
    int ret = JSDR_ERR;
    if (raw.alloc(SYMPBLOCK) == JSDR_OK && out.alloc(256) == JSDR_OK && drc.alloc(1) == JSDR_OK) {
        // the reference leaves RSdecdata untouched when RS fails: seed the device copy with the caller's bytes
        if (hipMemcpy(raw.p, raw_host, SYMPBLOCK, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(out.p, out_host, 256, hipMemcpyHostToDevice) == hipSuccess &&
            jsdr_fec_decode_batch(raw.p, 1, out.p, drc.p, 0) == JSDR_OK &&
            hipMemcpy(out_host, out.p, 256, hipMemcpyDeviceToHost) == hipSuccess &&
            hipMemcpy(rc, drc.p, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
            ret = JSDR_OK;
        else if (jsdr_last_error()[0] == 0)
            set_error("jsdr_fec_decode: transfer failed");
    }
    raw.release();
    out.release();
    drc.release();
    return ret;
}

int jsdr_fec_encode(const uint8_t data_host[256], uint8_t sym_host[5200])
{
    JSDR_REQUIRE(data_host && sym_host, "jsdr_fec_encode: null argument");
    DevBuf<unsigned char> data, sym;
    int ret = JSDR_ERR;
    if (data.alloc(256) == JSDR_OK && sym.alloc(SYMPBLOCK) == JSDR_OK) {
        if (hipMemcpy(data.p, data_host, 256, hipMemcpyHostToDevice) == hipSuccess &&
            jsdr_fec_encode_batch(data.p, 1, sym.p, 0) == JSDR_OK &&
            hipMemcpy(sym_host, sym.p, SYMPBLOCK, hipMemcpyDeviceToHost) == hipSuccess)
            ret = JSDR_OK;
        else if (jsdr_last_error()[0] == 0)
            set_error("jsdr_fec_encode: transfer failed");
    }
    data.release();
    sym.release();
    return ret;
}

}  // extern "C"

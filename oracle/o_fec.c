/*
 * o_fec.c -- oracle: AO-40 FEC decoder / encoder, restating FECDecoder.java.
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Integer only.
 *
 * Tables that have a closed form are GENERATED here from their definitions and pinned
 * against the reference's literal tables by digest (tests/golden/table_digests.json):
 *   ALPHA_TO / INDEX_OF  GF(2^8), field polynomial 0x187      (FECDecoder.java:145-181)
 *   Partab               8-bit parity                          (:40-57)
 *   Syms                 (P[i&0x4f]<<1) | (1-P[i&0x6d])        (:105-114, cf. :563-564)
 *   Scrambler            CCSDS randomiser x^8+x^7+x^5+x^3+1    (:118-139)
 *   RS_poly              generator of RS(255,223), FCR 112, PRIM 11, index form (:544-546)
 * mettab cannot be regenerated (two irregular entries): row 0 is data, row 1 is its
 * mirror with the two patches (:67-100).
 */
#include "jsdr_oracle.h"
#include <string.h>
#include <stdlib.h>

enum {
    NN = 255, KK = 223, NROOTS = 32, FCR = 112, PRIM = 11, IPRIM = 116, A0 = NN,
    BLOCKSIZE = 256, RSBLOCKS = 2, RSPAD = 95,
    VK = 7, CPOLYA = 0x4f, CPOLYB = 0x6d,
    NBITS = ((BLOCKSIZE + NROOTS * RSBLOCKS) * 8 + VK - 1), /* 2566 */
    ROWS = 80, COLUMNS = 65, SYMPBLOCK = ROWS * COLUMNS, SYNC_POLY = 0x48
};

static int tables_ready = 0;
static int ALPHA_TO[256], INDEX_OF[256], Partab[256], Syms[128], Scrambler[320], RS_poly[16];
static int mettab[2][256];

/* FECDecoder.java:67-83, row [0] (sent symbol 0) verbatim as data */
static const short mettab0[256] = {
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20, 20,
    20, 20, 20, 20, 20, 20, 20,
    19, 19, 19, 19, 19, 19, 19, 19, 19, 19, 19,
    18, 18, 18, 18, 18, 18, 17, 17, 17, 16, 16, 16, 15, 15, 14, 14, 13, 13, 12, 11,
    10, 10, 9, 8, 7, 6, 5, 3, 2, 1,
    -1, -2, -4, -5, -7, -9, -11, -13, -15, -17, -19, -21, -23, -25, -28, -30,
    -32, -35, -37, -40, -42, -45, -47, -50, -52, -55, -58, -60, -63, -66, -68, -71,
    -74, -77, -79, -82, -85, -88, -90, -93, -96, -99, -102, -104, -107, -110, -113, -116,
    -119, -121, -124, -127, -130, -133, -136, -138, -141, -144, -147, -150, -153, -155, -158, -161,
    -164, -167, -170, -172, -175, -178, -181, -184, -187, -190, -192, -195, -198, -201, -204, -207,
    -210, -212, -215, -218, -221, -224, -227, -229, -232, -235, -238, -241, -244, -247, -249, -252,
    -255, -258, -261, -264, -267, -269, -272, -275, -278, -281, -284, -286, -289, -292, -295, -298,
    -301, -304, -306, -309, -312, -315, -318, -320, -324, -326, -329, -332, -335, -337, -341, -372
};

static int parity8(int v)
{
    v ^= v >> 4;
    v ^= v >> 2;
    v ^= v >> 1;
    return v & 1;
}

static int mod255(int x) /* FECDecoder.java:317-323 */
{
    while (x >= 255) {
        x -= 255;
        x = (x >> 8) + (x & 255);
    }
    return x;
}

static void build_tables(void)
{
    if (tables_ready) return;
    /* GF(256), primitive polynomial x^8+x^7+x^2+x+1 (0x187) */
    int x = 1;
    for (int i = 0; i < 255; i++) {
        ALPHA_TO[i] = x;
        INDEX_OF[x] = i;
        x <<= 1;
        if (x & 0x100) x ^= 0x187;
    }
    ALPHA_TO[255] = 0;
    INDEX_OF[0] = A0;
    for (int i = 0; i < 256; i++) Partab[i] = parity8(i);
    for (int i = 0; i < 128; i++)
        Syms[i] = (Partab[i & CPOLYA] << 1) | (1 - Partab[i & CPOLYB]);
    /* CCSDS pseudo-randomiser: h(x)=x^8+x^7+x^5+x^3+1, all-ones start, MSB first */
    {
        unsigned sr = 0xff;
        for (int i = 0; i < 320; i++) {
            int byte = 0;
            for (int b = 0; b < 8; b++) {
                int out = sr & 1u;
                byte = (byte << 1) | out;
                unsigned fb = (sr ^ (sr >> 3) ^ (sr >> 5) ^ (sr >> 7)) & 1u;
                sr = (sr >> 1) | (fb << 7);
            }
            Scrambler[i] = byte;
        }
    }
    /* RS generator g(x) = prod_{i=0}^{31} (x - alpha^{PRIM*(FCR+i)}), poly form then index form;
     * RS_poly[j] = index form of g_{j+1}, j=0..15 (palindromic generator). */
    {
        int g[NROOTS + 1];
        memset(g, 0, sizeof(g));
        g[0] = 1;
        for (int i = 0; i < NROOTS; i++) {
            int root = mod255((FCR + i) * PRIM);
            g[i + 1] = 1;
            for (int j = i; j > 0; j--) {
                if (g[j] != 0)
                    g[j] = g[j - 1] ^ ALPHA_TO[mod255(INDEX_OF[g[j]] + root)];
                else
                    g[j] = g[j - 1];
            }
            g[0] = ALPHA_TO[mod255(INDEX_OF[g[0]] + root)];
        }
        for (int j = 0; j < 16; j++) RS_poly[j] = INDEX_OF[g[j + 1]];
    }
    for (int i = 0; i < 256; i++) {
        mettab[0][i] = mettab0[i];
        mettab[1][i] = mettab0[255 - i];
    }
    /* the two entries where row 1 is not the mirror of row 0 (FECDecoder.java:84) */
    mettab[1][2] = -338;
    mettab[1][8] = -321;
    tables_ready = 1;
}

int jo_fec_table(const char *which, int32_t *out, int cap)
{
    build_tables();
    const int *src = NULL;
    int n = 0;
    if (!strcmp(which, "ALPHA_TO")) { src = ALPHA_TO; n = 256; }
    else if (!strcmp(which, "INDEX_OF")) { src = INDEX_OF; n = 256; }
    else if (!strcmp(which, "Partab")) { src = Partab; n = 256; }
    else if (!strcmp(which, "Syms")) { src = Syms; n = 128; }
    else if (!strcmp(which, "Scrambler")) { src = Scrambler; n = 320; }
    else if (!strcmp(which, "RS_poly")) { src = RS_poly; n = 16; }
    else if (!strcmp(which, "mettab")) { src = &mettab[0][0]; n = 512; }
    else return -1;
    if (n > cap) n = cap;
    for (int i = 0; i < n; i++) out[i] = src[i];
    return n;
}

/* FECDecoder.java:203-278 viterbi27: K=7 r=1/2 soft decision, 64 states, path decisions
 * packed 32 per `long` element (two elements per decoded bit), chain-back from state 0. */
int jo_viterbi27(uint8_t *data, const uint8_t *symbols, int nbits)
{
    build_tables();
    int bitcnt = 0;
    int beststate, i, j, k = 0, l = 0;
    int64_t cmetric[64], nmetric[64];
    int64_t *pp = (int64_t *)calloc((size_t)nbits * 2, sizeof(int64_t));
    int64_t m0, m1, mask;
    int mets[4];

    cmetric[0] = 0;
    for (i = 1; i < 64; i++) cmetric[i] = -999999;

    for (;;) {
        for (i = 0; i < 4; i++) {
            mets[i] = 0;
            for (j = 0; j < 2; j++)
                mets[i] += mettab[(i >> (1 - j)) & 1][symbols[j + k] & 0xff];
        }
        k += 2;
        mask = 1;
        for (i = 0; i < 64; i += 2) {
            int b1, b2;
            b1 = mets[Syms[i]];
            nmetric[i] = m0 = cmetric[i / 2] + b1;
            b2 = mets[Syms[i + 1]];
            b1 -= b2;
            m1 = cmetric[(i / 2) + (1 << (VK - 2))] + b2;
            if (m1 > m0) {
                nmetric[i] = m1;
                pp[l] |= mask;
            }
            m0 -= b1;
            nmetric[i + 1] = m0;
            m1 += b1;
            if (m1 > m0) {
                nmetric[i + 1] = m1;
                pp[l] |= mask << 1;
            }
            mask <<= 2;
            if ((mask & 0xffffffffLL) == 0) {
                mask = 1;
                l++;
            }
        }
        if (mask != 1) l++;
        if (++bitcnt == nbits) {
            beststate = 0;
            break;
        }
        memcpy(cmetric, nmetric, sizeof(cmetric));
    }
    l -= 2;
    for (i = 0; i < nbits / 8; i++) data[i] = 0;
    for (i = nbits - VK; i >= 0; i--) {
        if ((pp[l + (beststate >> 5)] & ((int64_t)1 << (beststate & 31))) != 0) {
            beststate |= (1 << (VK - 1));
            data[i >> 3] |= (uint8_t)(0x80 >> (i & 7));
        }
        beststate >>= 1;
        l -= 2;
    }
    free(pp);
    return 0;
}

static int imin(int a, int b) { return a < b ? a : b; }

/* FECDecoder.java:325-519 decode_rs_8: syndromes, Berlekamp-Massey, Chien, Forney.
 * Deviation: `s[i] = data[0]` (:337) is unmasked in Java (sign-extends); masked here.  Column
 * 0 is RS padding (0) on every path through FECDecode, so the two agree there.           */
int jo_decode_rs_8(uint8_t *data, int *eras_pos, int no_eras)
{
    build_tables();
    int deg_lambda, el, deg_omega;
    int i, j, r, k;
    int u, q, tmp, num1, num2, den, discr_r;
    int lambda[NROOTS + 1], s[NROOTS];
    int b[NROOTS + 1], t[NROOTS + 1], omega[NROOTS + 1];
    int root[NROOTS], reg[NROOTS + 1], loc[NROOTS];
    int syn_error, count = -1;
    memset(lambda, 0, sizeof(lambda));
    memset(b, 0, sizeof(b));
    memset(t, 0, sizeof(t));
    memset(omega, 0, sizeof(omega));
    memset(root, 0, sizeof(root));
    memset(reg, 0, sizeof(reg));
    memset(loc, 0, sizeof(loc));

    for (i = 0; i < NROOTS; i++) s[i] = data[0] & 0xff;

    for (j = 1; j < NN; j++) {
        for (i = 0; i < NROOTS; i++) {
            if (s[i] == 0) {
                s[i] = data[j] & 0xff;
            } else {
                s[i] = (data[j] & 0xff) ^ ALPHA_TO[mod255(INDEX_OF[s[i]] + (FCR + i) * PRIM)];
            }
        }
    }

    syn_error = 0;
    for (i = 0; i < NROOTS; i++) {
        syn_error |= s[i];
        s[i] = INDEX_OF[s[i]];
    }

    if (0 == syn_error) {
        count = 0;
        goto finish;
    }
    lambda[0] = 1;

    if (no_eras > 0) {
        lambda[1] = ALPHA_TO[mod255(PRIM * (NN - 1 - eras_pos[0]))];
        for (i = 1; i < no_eras; i++) {
            u = mod255(PRIM * (NN - 1 - eras_pos[i]));
            for (j = i + 1; j > 0; j--) {
                tmp = INDEX_OF[lambda[j - 1]];
                if (tmp != A0) lambda[j] ^= ALPHA_TO[mod255(u + tmp)];
            }
        }
    }
    for (i = 0; i < NROOTS + 1; i++) b[i] = INDEX_OF[lambda[i]];

    r = no_eras;
    el = no_eras;
    while (++r <= NROOTS) {
        discr_r = 0;
        for (i = 0; i < r; i++) {
            if ((lambda[i] != 0) && (s[r - i - 1] != A0)) {
                discr_r ^= ALPHA_TO[mod255(INDEX_OF[lambda[i]] + s[r - i - 1])];
            }
        }
        discr_r = INDEX_OF[discr_r];
        if (discr_r == A0) {
            memmove(&b[1], b, NROOTS * sizeof(b[0]));
            b[0] = A0;
        } else {
            t[0] = lambda[0];
            for (i = 0; i < NROOTS; i++) {
                if (b[i] != A0)
                    t[i + 1] = lambda[i + 1] ^ ALPHA_TO[mod255(discr_r + b[i])];
                else
                    t[i + 1] = lambda[i + 1];
            }
            if (2 * el <= r + no_eras - 1) {
                el = r + no_eras - el;
                for (i = 0; i <= NROOTS; i++)
                    b[i] = (lambda[i] == 0) ? A0 : mod255(INDEX_OF[lambda[i]] - discr_r + NN);
            } else {
                memmove(&b[1], b, NROOTS * sizeof(b[0]));
                b[0] = A0;
            }
            memcpy(lambda, t, (NROOTS + 1) * sizeof(t[0]));
        }
    }

    deg_lambda = 0;
    for (i = 0; i < NROOTS + 1; i++) {
        lambda[i] = INDEX_OF[lambda[i]];
        if (lambda[i] != A0) deg_lambda = i;
    }
    memcpy(&reg[1], &lambda[1], NROOTS * sizeof(reg[0]));
    count = 0;
    for (i = 1, k = IPRIM - 1; i <= NN; i++, k = mod255(k + IPRIM)) {
        q = 1;
        for (j = deg_lambda; j > 0; j--) {
            if (reg[j] != A0) {
                reg[j] = mod255(reg[j] + j);
                q ^= ALPHA_TO[reg[j]];
            }
        }
        if (q != 0) continue;
        root[count] = i;
        loc[count] = k;
        if (++count == deg_lambda) break;
    }
    if (deg_lambda != count) {
        count = -1;
        goto finish;
    }
    deg_omega = 0;
    for (i = 0; i < NROOTS; i++) {
        tmp = 0;
        j = (deg_lambda < i) ? deg_lambda : i;
        for (; j >= 0; j--) {
            if ((s[i - j] != A0) && (lambda[j] != A0))
                tmp ^= ALPHA_TO[mod255(s[i - j] + lambda[j])];
        }
        if (tmp != 0) deg_omega = i;
        omega[i] = INDEX_OF[tmp];
    }
    omega[NROOTS] = A0;

    for (j = count - 1; j >= 0; j--) {
        num1 = 0;
        for (i = deg_omega; i >= 0; i--) {
            if (omega[i] != A0) num1 ^= ALPHA_TO[mod255(omega[i] + i * root[j])];
        }
        num2 = ALPHA_TO[mod255(root[j] * (FCR - 1) + NN)];
        den = 0;
        for (i = imin(deg_lambda, NROOTS - 1) & ~1; i >= 0; i -= 2) {
            if (lambda[i + 1] != A0) den ^= ALPHA_TO[mod255(lambda[i + 1] + i * root[j])];
        }
        if (den == 0) {
            count = -1;
            goto finish;
        }
        if (num1 != 0) {
            data[loc[j]] ^= (uint8_t)ALPHA_TO[mod255(INDEX_OF[num1] + INDEX_OF[num2] + NN - INDEX_OF[den])];
        }
    }
finish:
    if (eras_pos != NULL) {
        for (i = 0; i < count; i++) eras_pos[i] = loc[i];
    }
    return count;
}

/* ---- encoder (FECDecoder.java:538-688) ------------------------------------------ */
typedef struct {
    int Nbytes, Bindex, Conv_sr;
    int RS_block[RSBLOCKS][NROOTS];
    uint8_t *reencode; /* [SYMPBLOCK] */
} enc_t;

static void interleave_symbol(enc_t *e, int c) /* :549-556 */
{
    int row, col;
    col = e->Bindex / COLUMNS;
    row = e->Bindex % COLUMNS;
    if (c != 0) e->reencode[row * ROWS + col] = 1;
    e->Bindex++;
}

static void encode_and_interleave(enc_t *e, int c, int cnt) /* :559-566 */
{
    while (cnt-- != 0) {
        /* Java lets Conv_sr grow (int wraps); only bits 0..6 are ever read, keep 16 */
        e->Conv_sr = ((e->Conv_sr << 1) | (c >> 7)) & 0xffff;
        c <<= 1;
        interleave_symbol(e, Partab[e->Conv_sr & CPOLYA]);
        interleave_symbol(e, 1 - Partab[e->Conv_sr & CPOLYB]);
    }
}

static void scramble_and_encode(enc_t *e, int c) /* :569-572 */
{
    c ^= Scrambler[e->Nbytes];
    encode_and_interleave(e, c, 8);
}

static void local_init_encoder(enc_t *e) /* :584-606 */
{
    int i, j, sr;
    e->Nbytes = 0;
    e->Conv_sr = 0;
    e->Bindex = COLUMNS;
    for (j = 0; j < RSBLOCKS; j++)
        for (i = 0; i < NROOTS; i++) e->RS_block[j][i] = 0;
    for (i = 0; i < 5200; i++) e->reencode[i] = 0;
    sr = 0x7f;
    for (i = 0; i < 65; i++) {
        if ((sr & 64) != 0) e->reencode[ROWS * i] = 1;
        sr = (int)((unsigned)sr << 1) | Partab[sr & SYNC_POLY]; /* Java's int shift wraps; C's signed shift may not */
    }
}

static void local_encode_byte(enc_t *e, int c) /* :614-655 */
{
    int rsi, i, feedback;
    rsi = e->Nbytes & 1;
    feedback = INDEX_OF[c ^ e->RS_block[rsi][0]];
    if (feedback != A0) {
        for (int j = 0; j < 15; j++) {
            int t = ALPHA_TO[mod255(feedback + RS_poly[j])];
            e->RS_block[rsi][j + 1] ^= t;
            e->RS_block[rsi][31 - j] ^= t;
        }
        e->RS_block[rsi][16] ^= ALPHA_TO[mod255(feedback + RS_poly[15])];
    }
    for (i = 0; i < 31; i++) e->RS_block[rsi][i] = e->RS_block[rsi][i + 1];
    if (feedback != A0)
        e->RS_block[rsi][31] = ALPHA_TO[feedback];
    else
        e->RS_block[rsi][31] = 0;
    scramble_and_encode(e, c);
    e->Nbytes++;
}

static void local_encode_parity(enc_t *e) /* :662-671 */
{
    int c = e->RS_block[e->Nbytes & 1][(e->Nbytes - 256) >> 1];
    scramble_and_encode(e, c);
    if (++e->Nbytes == 320) encode_and_interleave(e, 0, 6);
}

/* NB: in Java `Conv_sr & CPOLYA` indexes Partab with a value < 256 because CPOLYx < 128. */
void jo_fec_encode(const uint8_t data[256], uint8_t sym[5200]) /* encode_FEC40 :677-688 */
{
    build_tables();
    enc_t e;
    e.reencode = sym;
    local_init_encoder(&e);
    for (int i = 0; i < 256; i++) local_encode_byte(&e, (int)data[i] & 0xff);
    for (int i = 0; i < 64; i++) local_encode_parity(&e);
}

/* FECDecoder.java:703-852 FECDecode: de-interleave, Viterbi, de-scramble + 2 x RS, re-encode
 * and count channel errors.  Returns -1 on RS failure, else the channel error count.      */
int jo_fec_decode(const uint8_t raw[5200], uint8_t RSdecdata[256])
{
    build_tables();
    uint8_t symbols[NBITS * 2 + 65 + 3];
    uint8_t vitdecdata[(NBITS - 6) / 8];
    int nRC = 0;
    memset(symbols, 0, sizeof(symbols));
    {
        int col, row, coltop = 0, rowstart;
        for (col = 1; col < ROWS; col++) {
            rowstart = 0;
            for (row = 0; row < COLUMNS; row++) {
                symbols[coltop + row] = raw[rowstart + col];
                rowstart += ROWS;
            }
            coltop += COLUMNS;
        }
    }
    jo_viterbi27(vitdecdata, symbols, NBITS);
    {
        uint8_t rsblocks[RSBLOCKS][NN];
        int row, col, di, si;
        int rserrs[RSBLOCKS];
        int rs_failures;
        memset(rsblocks, 0, sizeof(rsblocks));
        di = 0;
        si = 0;
        for (col = RSPAD; col < NN; col++) {
            for (row = 0; row < RSBLOCKS; row++) {
                rsblocks[row][col] = (uint8_t)(vitdecdata[di++] ^ Scrambler[si++]);
            }
        }
        rs_failures = 0;
        for (row = 0; row < RSBLOCKS; row++) {
            rserrs[row] = jo_decode_rs_8(rsblocks[row], NULL, 0);
            rs_failures += (rserrs[row] == -1) ? 1 : 0;
        }
        if (0 == rs_failures) {
            int j = 0;
            for (col = RSPAD; col < KK; col++) {
                for (row = 0; row < RSBLOCKS; row++) RSdecdata[j++] = rsblocks[row][col];
            }
        }
        for (row = 0; row < RSBLOCKS; row++) {
            if (rserrs[row] == -1) nRC = -1;
        }
    }
    if (nRC >= 0) {
        uint8_t reencode[SYMPBLOCK];
        int errors = 0;
        jo_fec_encode(RSdecdata, reencode);
        for (int i = 0; i < SYMPBLOCK; i++)
            if ((reencode[i] & 0xff) != ((raw[i] & 0xff) >> 7)) errors++;
        nRC = errors;
    }
    return nRC;
}

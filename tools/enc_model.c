// CPU model of the wide block matcher (zstd_encode.hip, HASH_LOG 13 variant) with switches for the pieces a
// higher effort tier could add, and a bit-cost estimate of the block each choice would produce.  Used to decide what to
// build before spending GPU time on it; it is a planning tool, not a codec (no bitstream is written).
//   gcc -O2 -o /tmp/enc_model tools/enc_model.c -lm && /tmp/enc_model text|binary [cap_MB]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <glob.h>
#include <sys/stat.h>

typedef struct {
    int hash_log, ways, lazy, rep, gate, custom_fse, huf_any, huf_opt, lmax_probe, mls, back;
} Opt;

static const int LLn[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const int MLn[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
static const int OFn[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const uint32_t LLbase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
static const uint8_t LLbits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const uint32_t MLbase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
static const uint8_t MLbits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};

static int ll_code(uint32_t ll) { int c = 35; while (LLbase[c] > ll) c--; return c; }
static int ml_code(uint32_t ml) { int c = 52; while (MLbase[c] > ml) c--; return c; }
static int hib(uint32_t v) { return 31 - __builtin_clz(v); }

typedef struct { uint32_t ll, ml, ov; } Seq;  // ov = offset value as coded (1..3 repeat codes, else offset + 3)

static double huf_cost_bits(const uint32_t *hist, uint32_t n, int optimal, int *nsym_out, int *maxsym_out) {
    // optimal: true Huffman lengths clamped to 11 with a Kraft repair; otherwise Shannon lengths + the same repair
    int len[256] = {0}, nsym = 0, maxsym = 0;
    for (int i = 0; i < 256; i++) if (hist[i]) { nsym++; maxsym = i; }
    *nsym_out = nsym; *maxsym_out = maxsym;
    if (nsym < 2) return 1e18;
    if (optimal == 1) {
        // plain O(n^2) Huffman
        static uint64_t w[512]; static int par[512]; static int alive[512];
        int cnt = 0, idx[256];
        for (int i = 0; i < 256; i++) if (hist[i]) { w[cnt] = hist[i]; alive[cnt] = 1; par[cnt] = -1; idx[cnt] = i; cnt++; }
        int total = cnt;
        for (int r = 0; r < nsym - 1; r++) {
            int a = -1, b = -1;
            for (int i = 0; i < total; i++) if (alive[i]) { if (a < 0 || w[i] < w[a]) { b = a; a = i; } else if (b < 0 || w[i] < w[b]) b = i; }
            w[total] = w[a] + w[b]; alive[total] = 1; par[total] = -1; alive[a] = alive[b] = 0; par[a] = par[b] = total; total++;
        }
        for (int i = 0; i < cnt; i++) { int d = 0, k = i; while (par[k] >= 0) { k = par[k]; d++; } len[idx[i]] = d > 11 ? 11 : d; }
    } else if (optimal == 2) {
        // rounded Shannon lengths, then repair by cost per unit of Kraft sum: lengthen where count << len is smallest, shorten where it is largest
        for (int i = 0; i < 256; i++) if (hist[i]) { double l = log2((double)n / hist[i]); int r = (int)(l + 0.5); len[i] = r < 1 ? 1 : (r > 11 ? 11 : r); }
        long K = 0; for (int i = 0; i < 256; i++) if (len[i]) K += 2048 >> len[i];
        while (K > 2048) { int best = -1; for (int i = 0; i < 256; i++) if (len[i] && len[i] < 11 && (best < 0 || ((uint64_t)hist[i] << len[i]) < ((uint64_t)hist[best] << len[best]))) best = i; if (best < 0) return 1e18; K -= 2048 >> len[best]; len[best]++; K += 2048 >> len[best]; }
        while (K < 2048) { long gap = 2048 - K; int best = -1; for (int i = 0; i < 256; i++) if (len[i] > 1 && (2048 >> len[i]) <= gap && (best < 0 || ((uint64_t)hist[i] << len[i]) > ((uint64_t)hist[best] << len[best]))) best = i; if (best < 0) break; K += 2048 >> len[best]; len[best]--; }
        double bits = 0; for (int i = 0; i < 256; i++) bits += (double)hist[i] * len[i];
        return bits;
    } else {
        for (int i = 0; i < 256; i++) if (hist[i]) { int l = 1; while (l < 11 && ((uint64_t)hist[i] << l) < n) l++; len[i] = l; }
    }
    // Kraft repair as in huf_literals
    long K = 0; for (int i = 0; i < 256; i++) if (len[i]) K += 2048 >> len[i];
    while (K > 2048) { int best = -1; for (int i = 0; i < 256; i++) if (len[i] && len[i] < 11 && (best < 0 || hist[i] < hist[best])) best = i; if (best < 0) return 1e18; K -= 2048 >> len[best]; len[best]++; K += 2048 >> len[best]; }
    while (K < 2048) { long gap = 2048 - K; int best = -1; for (int i = 0; i < 256; i++) if (len[i] > 1 && (2048 >> len[i]) <= gap && (best < 0 || hist[i] > hist[best])) best = i; if (best < 0) break; K += 2048 >> len[best]; len[best]--; }
    double bits = 0; for (int i = 0; i < 256; i++) bits += (double)hist[i] * len[i];
    return bits;
}

static uint32_t hsh(const uint8_t *p, const Opt *o) {
    uint64_t v; memcpy(&v, p, 8);
    if (o->mls == 4) return ((uint32_t)v * 2654435761u) >> (32 - o->hash_log);
    return (uint32_t)(((v << (64 - 8 * o->mls)) * 0x9E3779B185EBCA87ull) >> (64 - o->hash_log));
}
static double norm_cost(const uint32_t *h, int nsym, uint32_t total, int tl) {
    // bits of coding the histogram with counts normalised to 2^tl (every present symbol >= 1, remainder to the largest)
    int norm[64], big = 0; long sum = 0; const int ts = 1 << tl;
    for (int i = 0; i < nsym; i++) { norm[i] = 0; if (h[i]) { long v = ((long)h[i] * ts + total / 2) / total; norm[i] = v < 1 ? 1 : (int)v; sum += norm[i]; if (h[i] > h[big]) big = i; } }
    norm[big] += ts - sum;
    while (norm[big] < 1) { int b2 = -1; for (int i = 0; i < nsym; i++) if (norm[i] > 1 && (b2 < 0 || norm[i] > norm[b2])) b2 = i; if (b2 < 0) return 1e18; norm[b2]--; norm[big]++; }
    double bits = 0;
    for (int i = 0; i < nsym; i++) if (h[i]) bits -= h[i] * log2((double)norm[i] / ts);
    return bits;
}
static double block_cost(const uint8_t *in, uint32_t n, const Opt *o, double *lit_bytes, double *seq_bytes, uint32_t *nseq_out, uint32_t *nrep_out) {
    const uint32_t HS = 1u << o->hash_log;
    static int32_t *tab = NULL; static Seq *seqs = NULL; static uint8_t *lits = NULL;
    if (!tab) { tab = malloc(sizeof(int32_t) * 8 * (1 << 17)); seqs = malloc(sizeof(Seq) * 70000); lits = malloc(1 << 18); }
    for (uint32_t i = 0; i < HS * 8; i++) tab[i] = -1;
    uint32_t nseq = 0, nlit = 0, anchor = 0, base = 0, nrep = 0;
    const uint32_t scan_end = n >= 8 ? n - 7 : 0;
    uint32_t rep[3] = {0, 0, 0};  // 0 = unknown at block start (blocks are encoded independently)
    uint32_t cand[64], mlen[64], crep[64];
    while (base < scan_end) {
        // all lanes probe with the table state of the window's start, then all insert
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t pos = base + l; cand[l] = 0; mlen[l] = 0; crep[l] = 0;
            if (pos >= scan_end) continue;
            uint32_t v; memcpy(&v, in + pos, 4);
            const uint32_t h = hsh(in + pos, o);
            uint32_t best = 0, bc = 0;
            for (int w = 0; w < o->ways; w++) {
                const int32_t c = tab[h * 8 + w];
                if (c < 0) continue;
                uint32_t cv; memcpy(&cv, in + c, 4);
                if (cv != v) continue;
                uint32_t k = 4; const uint32_t lim = n - pos < (uint32_t)o->lmax_probe ? n - pos : (uint32_t)o->lmax_probe;
                while (k < lim && in[pos + k] == in[c + k]) k++;
                if (k > best) { best = k; bc = (uint32_t)c; }
            }
            cand[l] = bc; mlen[l] = best;
        }
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t pos = base + l;
            if (pos >= scan_end) continue;
            const uint32_t h = hsh(in + pos, o);
            for (int w = o->ways - 1; w > 0; w--) tab[h * 8 + w] = tab[h * 8 + w - 1];
            tab[h * 8] = (int32_t)pos;
        }
        // pick left to right
        for (;;) {
            uint32_t l = anchor > base ? anchor - base : 0;
            // repeat-offset probe at every lane from the anchor on (rep[0] of the moment), when enabled
            uint32_t win = 64;
            for (; l < 64; l++) {
                const uint32_t pos = base + l;
                if (pos >= scan_end) break;
                uint32_t rl = 0;
                if (o->rep && rep[0] && pos >= rep[0]) {
                    uint32_t a, b; memcpy(&a, in + pos, 4); memcpy(&b, in + pos - rep[0], 4);
                    if (a == b) { rl = 4; while (pos + rl < n && in[pos + rl] == in[pos - rep[0] + rl]) rl++; }
                }
                if (o->rep >= 2) for (int ri = 1; ri < 3; ri++) if (rep[ri] && pos >= rep[ri]) {  // older history entries as ordinary candidates
                    uint32_t a, b; memcpy(&a, in + pos, 4); memcpy(&b, in + pos - rep[ri], 4);
                    if (a == b) { uint32_t k = 4; while (pos + k < n && k < 64 && in[pos + k] == in[pos - rep[ri] + k]) k++; if (k + 1 >= mlen[l] && k > rl) { mlen[l] = k; cand[l] = pos - rep[ri]; } }
                }
                crep[l] = rl;
                uint32_t m = mlen[l];
                if (m && o->gate) {  // cost gate: short far matches lose to literals
                    const uint32_t off = pos - cand[l];
                    if (o->gate == 1 && ((m == 4 && off > 2048) || (m == 5 && off > 32768))) m = 0;
                    if (o->gate == 2 && ((m == 4 && off > 256) || (m == 5 && off > 4096) || (m == 6 && off > 65536))) m = 0;
                    if (o->gate == 3 && ((m == 4) || (m == 5 && off > 1024) || (m == 6 && off > 16384))) m = 0;
                }
                if (rl && rl + 1 >= m) { win = l; break; }
                if (m) { crep[l] = 0; win = l; mlen[l] = m; break; }
            }
            if (win == 64) break;
            uint32_t pick = win;
            if (o->lazy && !crep[win]) {
                for (int step = 1; step <= o->lazy; step++) {
                    const uint32_t nx = pick + 1;
                    if (nx < 64 && base + nx < scan_end && mlen[nx] > mlen[pick] + 1) pick = nx; else break;
                }
            }
            uint32_t mpos = base + pick;
            uint32_t ml, off; int isrep = 0;
            if (crep[pick] && pick == win) { ml = crep[pick]; off = rep[0]; isrep = 1; }
            else { ml = mlen[pick]; off = mpos - cand[pick]; if (ml >= (uint32_t)o->lmax_probe) while (mpos + ml < n && in[mpos + ml] == in[mpos - off + ml]) ml++; }
            if (o->back) while (mpos > anchor && mpos > off && in[mpos - 1] == in[mpos - off - 1]) { mpos--; ml++; }
            const uint32_t ll = mpos - anchor;
            memcpy(lits + nlit, in + anchor, ll); nlit += ll;
            Seq s; s.ll = ll; s.ml = ml;
            if (isrep && ll != 0) { s.ov = 1; nrep++; }
            else if (o->rep && !isrep && rep[1] && off == rep[1] && ll != 0) { s.ov = 2; uint32_t t = rep[1]; rep[1] = rep[0]; rep[0] = t; nrep++; }
            else if (o->rep && !isrep && rep[0] && off == rep[0] && ll != 0) { s.ov = 1; nrep++; }
            else if (isrep && ll == 0) {  // rep[0] with ll == 0 is not expressible (code 1 means rep[1] then): explicit offset
                s.ov = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off;
            } else { s.ov = off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = off; }
            seqs[nseq++] = s;
            anchor = mpos + ml;
        }
        base = anchor > base + 64 ? anchor : base + 64;
    }
    memcpy(lits + nlit, in + anchor, n - anchor); nlit += n - anchor;
    // ---- cost ----
    uint32_t hist[256] = {0};
    for (uint32_t i = 0; i < nlit; i++) hist[lits[i]]++;
    int nsym, maxsym;
    double lb = nlit + 3;
    if (nlit >= 256) {
        double hb = huf_cost_bits(hist, nlit, o->huf_opt, &nsym, &maxsym);
        double tree = maxsym < 128 ? 1 + (maxsym + 1) / 2 : (o->huf_any ? 1 + nsym * 0.45 + 8 : 1e18);
        double tot = hb / 8 + tree + 6 + 4 + 5;
        if (tot < lb) lb = tot;
    }
    double sb = 0;
    if (nseq) {
        uint32_t hl[36] = {0}, hm[53] = {0}, ho[32] = {0};
        double extra = 0;
        for (uint32_t i = 0; i < nseq; i++) {
            const int lc = ll_code(seqs[i].ll), mc = ml_code(seqs[i].ml), oc = hib(seqs[i].ov);
            hl[lc]++; hm[mc]++; ho[oc]++;
            extra += LLbits[lc] + MLbits[mc] + oc;
        }
        double code_bits = 0;
        if (o->custom_fse >= 2) {
            const int tl = o->custom_fse == 2 ? 6 : (o->custom_fse == 3 ? 9 : o->custom_fse);
            int present = 0;
            for (int i = 0; i < 36; i++) present += hl[i] != 0;
            for (int i = 0; i < 53; i++) present += hm[i] != 0;
            for (int i = 0; i < 32; i++) present += ho[i] != 0;
            code_bits = norm_cost(hl, 36, nseq, tl) + norm_cost(hm, 53, nseq, tl) + norm_cost(ho, 32, nseq, tl == 9 ? 8 : tl) + present * 5.5 + 24;
        } else if (o->custom_fse) {
            int present = 0;
            for (int i = 0; i < 36; i++) if (hl[i]) { code_bits -= hl[i] * log2((double)hl[i] / nseq); present++; }
            for (int i = 0; i < 53; i++) if (hm[i]) { code_bits -= hm[i] * log2((double)hm[i] / nseq); present++; }
            for (int i = 0; i < 32; i++) if (ho[i]) { code_bits -= ho[i] * log2((double)ho[i] / nseq); present++; }
            code_bits = code_bits * 1.01 + present * 5.5 + 24;
        } else {
            for (int i = 0; i < 36; i++) if (hl[i]) code_bits -= hl[i] * log2((LLn[i] < 0 ? 1 : LLn[i]) / 64.0);
            for (int i = 0; i < 53; i++) if (hm[i]) code_bits -= hm[i] * log2((MLn[i] < 0 ? 1 : MLn[i]) / 64.0);
            for (int i = 0; i < 29; i++) if (ho[i]) code_bits -= ho[i] * log2((OFn[i] < 0 ? 1 : OFn[i]) / 32.0);
        }
        sb = (code_bits + extra) / 8 + 4;
    } else sb = 1;
    *lit_bytes = lb; *seq_bytes = sb; *nseq_out = nseq; *nrep_out = nrep;
    double tot = 3 + lb + sb;
    if (tot >= n + 3) { tot = n + 3; }
    return tot;
}

int main(int argc, char **argv) {
    const char *kind = argc > 1 ? argv[1] : "text";
    const double cap = (argc > 2 ? atof(argv[2]) : 32) * 1e6;
    const char *pt[] = {"/usr/lib/python3.10/*.py", "/usr/lib/python3.10/*/*.py", "/usr/lib/python3.10/*/*/*.py", "/usr/lib/python3/dist-packages/*/*.py", "/usr/lib/python3/dist-packages/*/*/*.py", NULL};
    const char *pb[] = {"/opt/rocm/lib/*.so*", "/usr/lib/x86_64-linux-gnu/*.so*", NULL};
    const char **pats = strcmp(kind, "text") ? pb : pt;
    glob_t g; memset(&g, 0, sizeof g);
    for (int i = 0; pats[i]; i++) glob(pats[i], i ? GLOB_APPEND : 0, NULL, &g);
    Opt opts[16]; int NO = 0;
    for (int i = 3; i < argc && NO < 16; i++) {
        Opt *q = &opts[NO++]; q->back = 0;
        sscanf(argv[i], "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", &q->hash_log, &q->ways, &q->lazy, &q->rep, &q->gate, &q->custom_fse, &q->huf_any, &q->huf_opt, &q->lmax_probe, &q->mls, &q->back);
    }
    double tot_in = 0, out[16] = {0}, lit[16] = {0}, seq[16] = {0}; uint64_t ns[16] = {0}, nr[16] = {0};
    for (size_t f = 0; f < g.gl_pathc && tot_in < cap; f++) {
        struct stat st;
        if (lstat(g.gl_pathv[f], &st) || !S_ISREG(st.st_mode) || st.st_size == 0) continue;
        FILE *fp = fopen(g.gl_pathv[f], "rb"); if (!fp) continue;
        size_t want = st.st_size; if (tot_in + want > cap) want = (size_t)(cap - tot_in) + 1; uint8_t *b = malloc(want + 64); size_t got = fread(b, 1, want, fp); fclose(fp);
        memset(b + got, 0, 64);
        for (size_t r = 0; r < got; r += 8u << 20) {  // rounds of 8 MiB, blocks of 128 KiB
            const size_t rl = got - r < (8u << 20) ? got - r : (8u << 20);
            for (size_t o = 0; o < rl; o += 131072) {
                const uint32_t n = rl - o < 131072 ? (uint32_t)(rl - o) : 131072u;
                for (int k = 0; k < NO; k++) {
                    double lbv, sbv; uint32_t q, rr;
                    out[k] += block_cost(b + r + o, n, &opts[k], &lbv, &sbv, &q, &rr);
                    lit[k] += lbv; seq[k] += sbv; ns[k] += q; nr[k] += rr;
                }
            }
            for (int k = 0; k < NO; k++) out[k] += 6;  // frame header
        }
        tot_in += got; free(b);
    }
    printf("%s: %.1f MB\n", kind, tot_in / 1e6);
    for (int k = 0; k < NO; k++)
        printf("hl%d w%d lazy%d rep%d gate%d fse%d hufany%d hufopt%d mls%d back%d: ratio %.4f  (lit %.4f seq %.4f) nseq %lu rep %.1f%%\n", opts[k].hash_log, opts[k].ways, opts[k].lazy,
               opts[k].rep, opts[k].gate, opts[k].custom_fse, opts[k].huf_any, opts[k].huf_opt, opts[k].mls, opts[k].back, out[k] / tot_in, lit[k] / tot_in, seq[k] / tot_in, (unsigned long)ns[k], 100.0 * nr[k] / (ns[k] ? ns[k] : 1));
    return 0;
}

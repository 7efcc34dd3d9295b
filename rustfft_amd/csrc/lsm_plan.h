// Host planner of the LDS stage machine (lsm.h): turns a length into the tree the reference's planner would build for it
// (src/plan.rs:412-425 design_fft_with_factors -> MixedRadix over the smooth part and Rader per large prime factor, :636-665
// design_prime -> Rader whose inner FFT of length q - 1 is planned recursively), and compiles the tree into a PROGRAM of in-place
// butterfly stages, two permutation tables and the stage tables.  No HIP here: tests drive it on the CPU.
//
// Nodes.  Every node owns a range of `phys` slots (relative positions 0 .. phys - 1, placed at base + sigma * position by its
// parent) and has an INPUT layout in_pos[i] and an OUTPUT layout out_pos[k] (slot of logical element i / k).  Every node runs in
// two modes that compute the same DFT: FWD takes in_pos -> out_pos, TRN (the transposed flow graph; the DFT matrix is symmetric)
// takes out_pos -> in_pos.  That is what lets Rader's second inner transform start from where the first one left its spectrum,
// without a permutation pass in between:
//   LEAF(L = r_0 .. r_{P-1})   FWD = decimation in time (digit-reversed in, natural out), TRN = decimation in frequency.
//   MIXED(A, B), N = |A| |B|   six-step (mixed_radix.rs:128-158) with A along the rows (stride 1) and B down the columns:
//                              FWD: A.FWD on every row, twiddle w_N^(n2 k1), B.FWD on every column; input n = |B| n1 + n2 at
//                              (B.in[n2], A.in[n1]), output k = k1 + |A| k2 at (B.out[k2], A.out[k1]).  TRN: B.TRN, twiddle, A.TRN.
//   RADER(q, I), |I| = q - 1   x[0] in the spare slot I.phys, x[g^(j+1)] at I.in[j];  I.FWD;  FIX;  multiply by D;  I.TRN;
//                              X[0] in the spare slot, X[g^(j-1)] at I.in[j].  No conjugates: with u[j] = x[g^(j+a)],
//                              v[i] = w_q^(g^-i), D = DFT(v) / (q - 1):  DFT(DFT(u) D)[i] = (u * v)[-i] = X[g^(i-a)] - x[0]
//                              (raders_algorithm.rs:235-283 computes the same convolution through conj(FFT(conj .))).  The same
//                              stages read as TRN: x[g^(j-1)] in, X[g^(j+1)] out -- D does not depend on the offset a.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "lsm.h"

namespace mi355 {
namespace lsm {

typedef std::complex<double> cd;

struct Hooks {
    std::function<cd(size_t index, size_t len)> tw;  // forward twiddle exp(-2 pi i index / len) (src/twiddles.rs:6-23)
    std::function<void(std::vector<cd>&)> dft;       // forward DFT in f64, any length (host side, tables only)
};

enum NodeKind { LEAF = 0, RADER = 1, MIXED = 2 };
enum Mode { FWD = 0, TRN = 1 };

struct Node {
    int kind = LEAF, len = 1, phys = 1;
    std::vector<int> radices;           // LEAF
    int q = 0, g = 0;                   // RADER
    std::unique_ptr<Node> inner, a, b;  // RADER: inner; MIXED: a (rows), b (columns)
    std::vector<int> in_pos, out_pos;
};

// a stage as the emitter describes it: the mixed-radix digits of the work-item index with their slot / twiddle-row / pre-multiplier strides
// (host only: finish() expands it into the kernel's descriptor words)
struct HostStage {
    int op = 0, radix = 1, flags = 0, items = 1;
    std::vector<int> n, a, t, p;  // digit 0 fastest
    std::vector<int> inner;       // 1: a digit of the transform itself (its value is 0 in the item that holds element 0), 0: a batch digit
    bool fix = false;             // BFLY2: the item whose inner digits are all 0 does the x[0] step
    int fix_off = 0, p2_delta = 0, fix_s0 = 0;  // fix_s0: in-row slot of S[0] relative to the instance (the item whose base it is does the x[0] step)
    int abase = 0, astep = 0, tw_off = 0, tw_kstep = 0, p_off = 0, p_kstep = 0;
    float cfix = 0;
};
struct Program {
    int n = 0, f = 1, rp = 0, nt = 256, tab_off = 0;
    size_t lds_elems = 0;  // rows + tables
    std::vector<HostStage> hstages;
    std::vector<LsmStage> stages;
    std::vector<unsigned> desc;
    std::vector<cd> ltab, gtab;
    std::vector<unsigned short> ldperm, stperm;
    std::string desc_str;
    int root_kind = LEAF;
    bool order_items = true;  // lane order of the work items by LDS bank (measurement builds switch it off)
    double est = 0;  // the cost model's SIMD time per row (arbitrary units): picks rows per workgroup and the block size
};

inline unsigned rcp32(int n) { return n <= 1 ? 0u : (unsigned)((((unsigned long long)1 << 32) + (unsigned)n - 1) / (unsigned)n); }
inline unsigned long long modpow(unsigned long long b, unsigned long long e, unsigned long long m) {
    unsigned long long r = 1;
    b %= m;
    while (e) {
        if (e & 1) r = r * b % m;
        b = b * b % m;
        e >>= 1;
    }
    return r;
}
inline std::vector<int> prime_factors(int n) {
    std::vector<int> f;
    for (int d = 2; (long long)d * d <= n; ++d)
        while (n % d == 0) {
            f.push_back(d);
            n /= d;
        }
    if (n > 1) f.push_back(n);
    return f;
}
// smallest primitive root (src/math_utils.rs:3-20)
inline int primitive_root(int p) {
    std::vector<int> fs = prime_factors(p - 1);
    fs.erase(std::unique(fs.begin(), fs.end()), fs.end());
    for (int g = 2; g < p; ++g) {
        bool ok = true;
        for (int f : fs)
            if (modpow(g, (p - 1) / f, p) == 1) {
                ok = false;
                break;
            }
        if (ok) return g;
    }
    return 0;
}
// the kernel's radix set: 2 .. 16, i.e. prime factors up to 13; larger prime factors are Rader nodes (17 - 1 = 16, 19 - 1 = 18 ... are smooth).
// (Measured and dropped, round 6: the primes 17 .. 61 as single in-register stages -- the conjugate-symmetric prime DFT streamed pair by
// pair -- compile to 207 .. 333 VGPRs whatever fences and register walls are put in, run one wave per SIMD on dependent multiply-add chains
// and reach 0.4 - 1.0 TB/s where the Rader form below reaches 1.9 - 2.7: profiles/r6/lsm_prime_stage_ab.txt.)
inline bool radix_ok(int r, int pmax) { return r >= 2 && r <= 16 && pmax >= 2; }
inline int items_per_thread(int r) { return std::min(kLsmEmax / r, kLsmItems); }
// What the kernel variant offers: nt threads with emax values each; `row` = the length of the whole transform (a stage touches at
// most that many elements of a row, i.e. runs at most row / r butterflies per row)
struct Ctx {
    int pmax = 13, nt = 256, emax = 16, row = 0;  // pmax: the largest prime radix of the kernel variant
    // a radix-r stage over the whole row fits the registers of one workgroup row: ceil((row / r) / nt) <= items a thread may hold
    bool fits(int) const { return true; }  // (a stage with more items than the workgroup holds at once runs in rounds: lsm.h)
};
// rough VALU cost of one radix-r butterfly with its twiddles and slot arithmetic (instructions per work item; planning only)
inline double item_cost(int r) {
    double lg = 0;
    for (int v = 1; v < r; v *= 2) lg += 1;
    return 8.0 * r + 2.5 * r * lg + 12.0;
}
// Radices of a leaf: the fewest stages; among those the factorisation whose most demanding stage needs the fewest threads per instance
// ((L / r) / items a thread may hold: that bounds the rows a workgroup can take), then the cheapest; large radices first (small strides get
// the big radices: the smallest twiddle tables)
inline bool choose_radices(int L, const Ctx& cx, std::vector<int>& out) {
    const int pmax = cx.pmax;
    out.clear();
    if (L == 1) return true;
    for (int f : prime_factors(L))
        if (f > pmax) return false;
    std::map<int, int> best;  // remaining -> stages
    std::function<int(int)> cost = [&](int rem) -> int {
        if (rem == 1) return 0;
        auto it = best.find(rem);
        if (it != best.end()) return it->second;
        int c = 1 << 20;
        for (int r = std::max(pmax, 16); r >= 2; --r)
            if (radix_ok(r, pmax) && cx.fits(r) && rem % r == 0) c = std::min(c, 1 + cost(rem / r));
        best[rem] = c;
        return c;
    };
    if (cost(L) >= (1 << 20)) return false;
    std::vector<int> cur, pick;
    double pick_demand = 1e30, pick_cost = 1e30;
    std::function<void(int, int)> walk = [&](int rem, int max_r) {  // non-increasing radices on optimal paths
        if (rem == 1) {
            double demand = 0, c = 0;
            for (int r : cur) {
                demand = std::max(demand, (double)(L / r) / items_per_thread(r));
                c += item_cost(r) * (L / r);
            }
            if (demand < pick_demand - 1e-9 || (demand < pick_demand + 1e-9 && c < pick_cost)) {
                pick = cur;
                pick_demand = demand;
                pick_cost = c;
            }
            return;
        }
        for (int r = std::min(max_r, std::max(pmax, 16)); r >= 2; --r)
            if (radix_ok(r, pmax) && cx.fits(r) && rem % r == 0 && cost(rem / r) == cost(rem) - 1) {
                cur.push_back(r);
                walk(rem / r, r);
                cur.pop_back();
            }
    };
    walk(L, std::max(pmax, 16));
    if (pick.empty()) return false;
    out = pick;
    return true;
}

inline std::unique_ptr<Node> make_leaf(int L, const Ctx& cx) {
    std::unique_ptr<Node> n(new Node());
    n->kind = LEAF;
    n->len = n->phys = L;
    if (!choose_radices(L, cx, n->radices)) return nullptr;
    // digit reversal of the in-place decimation in time: pos_P(i) = (i mod r_{P-1}) L / r_{P-1} + pos_{P-1}(i div r_{P-1})
    n->in_pos.assign(L, 0);
    n->out_pos.resize(L);
    for (int i = 0; i < L; ++i) {
        int pos = 0, rem = i, len = L;
        for (int p = (int)n->radices.size() - 1; p >= 0; --p) {
            const int r = n->radices[p];
            len /= r;
            pos += (rem % r) * len;
            rem /= r;
        }
        n->in_pos[i] = pos;
        n->out_pos[i] = i;
    }
    return n;
}
std::unique_ptr<Node> make_node(int N, const Ctx& cx, int depth);
inline std::unique_ptr<Node> make_rader(int q, const Ctx& cx, int depth) {
    std::unique_ptr<Node> n(new Node());
    n->kind = RADER;
    n->len = q;
    n->q = q;
    n->g = primitive_root(q);
    n->inner = make_node(q - 1, cx, depth + 1);
    if (!n->inner || !n->g) return nullptr;
    const Node& I = *n->inner;
    n->phys = I.phys + 1;
    n->in_pos.assign(q, 0);
    n->out_pos.assign(q, 0);
    n->in_pos[0] = n->out_pos[0] = I.phys;
    for (int j = 0; j < q - 1; ++j) {
        n->in_pos[(int)modpow(n->g, (unsigned)(j + 1), q)] = I.in_pos[j];
        n->out_pos[(int)modpow(n->g, (unsigned)((j + q - 2) % (q - 1)), q)] = I.in_pos[j];  // X[g^(j-1)]
    }
    return n;
}
inline std::unique_ptr<Node> make_mixed(std::unique_ptr<Node> a, std::unique_ptr<Node> b) {
    std::unique_ptr<Node> n(new Node());
    n->kind = MIXED;
    const int la = a->len, lb = b->len, pa = a->phys;
    n->len = la * lb;
    n->phys = b->phys * pa;
    n->in_pos.resize(n->len);
    n->out_pos.resize(n->len);
    for (int n1 = 0; n1 < la; ++n1)
        for (int n2 = 0; n2 < lb; ++n2) {
            n->in_pos[lb * n1 + n2] = b->in_pos[n2] * pa + a->in_pos[n1];
            n->out_pos[n1 + la * n2] = b->out_pos[n2] * pa + a->out_pos[n1];
        }
    n->a = std::move(a);
    n->b = std::move(b);
    return n;
}
// the reference's tree for a length: the smooth part as one LEAF, one RADER per prime factor outside the radix set, combined by
// MIXED nodes (the Rader factors along the rows)
inline std::unique_ptr<Node> make_node(int N, const Ctx& cx, int depth) {
    if (depth > 3 || N < 2) return nullptr;
    int smooth = 1;
    std::vector<int> large;
    for (int f : prime_factors(N)) {
        if (f <= cx.pmax)
            smooth *= f;
        else
            large.push_back(f);
    }
    std::unique_ptr<Node> node;
    if (smooth > 1) {
        node = make_leaf(smooth, cx);
        if (!node) return nullptr;
    }
    for (int q : large) {
        std::unique_ptr<Node> r = make_rader(q, cx, depth);
        if (!r) return nullptr;
        node = node ? make_mixed(std::move(r), std::move(node)) : std::move(r);
    }
    return node;
}
inline std::string describe(const Node& n) {
    if (n.kind == LEAF) {
        std::string s = "leaf" + std::to_string(n.len) + "(";
        for (size_t i = 0; i < n.radices.size(); ++i) s += (i ? "x" : "") + std::to_string(n.radices[i]);
        return s + ")";
    }
    if (n.kind == RADER) return "rader" + std::to_string(n.q) + "[" + describe(*n.inner) + "]";
    return "mixed{" + describe(*n.a) + ", " + describe(*n.b) + "}";
}

// ---- emission --------------------------------------------------------------------------------------------------------------
struct Dim {
    int n, stride;
};
struct Frame {
    int base = 0, sigma = 1;
    std::vector<Dim> batch;  // outermost first
};
struct PreOp {
    bool on = false, global = false;
    int tab_off = 0, base = 0, sigma = 1, n_outer = 0;  // table index of slot x = tab_off + (x - base) / sigma over the dims inside
};
struct Emitter {
    Program* prog;
    const Hooks* hooks;
    bool tw_global = false;
    bool fuse = true;  // Rader's middle stages as one double-butterfly stage (measurement builds switch it off)
    bool failed = false;
    std::map<std::pair<int, int>, int> leaf_tw;  // (s r, r) -> offset in ltab
    std::map<int, int> d_tab;                    // q -> offset (the inner layout is the same for every instance of a prime)
    std::map<const Node*, int> mixed_tab;        // six-step table of a MIXED node (FWD and TRN use the same one)
    int n_mixed = 0;

    struct Dig {
        int n, a, t, p;
        int inner = 0;
    };
    void push_stage(int op, int radix, int flags, int abase, int astep, std::vector<Dig> digs, int tw_off, int tw_kstep, const PreOp& pre, int p_abs_off,
                    int p_kstep, float cfix) {
        // drop unit digits, smallest slot stride fastest across lanes, merge what is one linear run
        std::vector<Dig> d;
        for (auto& x : digs)
            if (x.n > 1) d.push_back(x);
        std::stable_sort(d.begin(), d.end(), [](const Dig& x, const Dig& y) { return x.a < y.a; });
        for (size_t i = 0; i + 1 < d.size();) {
            if (d[i + 1].a == d[i].n * d[i].a && d[i + 1].t == d[i].n * d[i].t && d[i + 1].p == d[i].n * d[i].p && d[i + 1].inner == d[i].inner) {
                d[i].n *= d[i + 1].n;
                d.erase(d.begin() + i + 1);
            } else {
                ++i;
            }
        }
        HostStage st;
        st.op = op;
        st.radix = radix;
        st.flags = flags | (pre.on ? (LSM_PRE_MUL | (pre.global ? LSM_PRE_GLOBAL : 0)) : 0);
        for (size_t i = 0; i < d.size(); ++i) {
            st.n.push_back(d[i].n);
            st.a.push_back(d[i].a);
            st.t.push_back(d[i].t);
            st.p.push_back(d[i].p);
            st.inner.push_back(d[i].inner);
            st.items *= d[i].n;
        }
        st.abase = abase;
        st.astep = astep;
        st.tw_off = tw_off;
        st.tw_kstep = tw_kstep;
        st.p_off = p_abs_off;
        st.p_kstep = p_kstep;
        st.cfix = cfix;
        prog->hstages.push_back(st);
    }
    // pre-multiplier strides of a digit with slot stride a: inside the table's owner -> a / sigma, outside -> 0
    int pstride(const PreOp& pre, int a, bool outer) const { return (!pre.on || outer) ? 0 : a / pre.sigma; }

    int leaf_table(int s, int r) {
        auto key = std::make_pair(s * r, r);
        auto it = leaf_tw.find(key);
        if (it != leaf_tw.end()) return it->second;
        const int off = (int)prog->ltab.size();
        for (int k = 1; k < r; ++k)
            for (int lo = 0; lo < s; ++lo) prog->ltab.push_back(hooks->tw((size_t)lo * k, (size_t)s * r));
        leaf_tw[key] = off;
        return off;
    }
    void emit_leaf(const Node& nd, int mode, const Frame& fr, const PreOp& pre) {
        const int P = (int)nd.radices.size(), L = nd.len;
        std::vector<int> s(P);
        int acc = 1;
        for (int p = 0; p < P; ++p) {
            s[p] = acc;
            acc *= nd.radices[p];
        }
        bool first = true;
        for (int i = 0; i < P; ++i) {
            const int p = mode == FWD ? i : P - 1 - i, r = nd.radices[p], sp = s[p];
            std::vector<Dig> digs;
            const PreOp use = first ? pre : PreOp{};
            digs.push_back(Dig{sp, fr.sigma, 1, pstride(use, fr.sigma, false), 1});
            digs.push_back(Dig{L / (sp * r), fr.sigma * sp * r, 0, pstride(use, fr.sigma * sp * r, false), 1});
            for (size_t b = 0; b < fr.batch.size(); ++b) digs.push_back(Dig{fr.batch[b].n, fr.batch[b].stride, 0, pstride(use, fr.batch[b].stride, (int)b < use.n_outer)});
            const int flags = sp > 1 ? (mode == FWD ? LSM_TW_PRE : LSM_TW_POST) : 0;
            const int astep = fr.sigma * sp;
            push_stage(LSM_BFLY, r, flags, fr.base, astep, digs, sp > 1 ? leaf_table(sp, r) : 0, sp, use, use.on ? use.tab_off + (fr.base - use.base) / use.sigma : 0,
                       use.on ? astep / use.sigma : 0, 0.f);
            first = false;
        }
    }
    void emit_rader(const Node& nd, const Frame& fr, const PreOp& pre) {
        const Node& I = *nd.inner;
        const int q = nd.q, L = q - 1;
        emit(I, FWD, fr, pre);
        const size_t last1 = prog->hstages.size() - 1;
        int doff;
        auto it = d_tab.find(q);
        if (it != d_tab.end()) {
            doff = it->second;
        } else {
            // D = DFT(v) / L, v[i] = w_q^(g^-i), stored by the slot the first inner transform leaves bin j in
            std::vector<cd> v(L);
            const unsigned long long ginv = modpow(nd.g, (unsigned)(q - 2), q);
            unsigned long long e = 1;
            for (int i = 0; i < L; ++i) {
                v[i] = hooks->tw((size_t)e, (size_t)q);
                e = e * ginv % q;
            }
            hooks->dft(v);
            doff = (int)prog->ltab.size();
            prog->ltab.resize(prog->ltab.size() + I.phys, cd(0, 0));
            for (int j = 0; j < L; ++j) prog->ltab[doff + I.out_pos[j]] = v[j] / (double)L;
            d_tab[q] = doff;
        }
        PreOp pd;
        pd.on = true;
        pd.global = false;
        pd.tab_off = doff;
        pd.base = fr.base;
        pd.sigma = fr.sigma;
        pd.n_outer = (int)fr.batch.size();
        const int s0 = fr.base + fr.sigma * I.out_pos[0], x0 = fr.base + fr.sigma * I.phys;
        // The last stage of the first inner transform and the first stage of the second one work on the same slots (the transposed flow
        // graph starts where the forward one ended): emitted apart, then FUSED into one double-butterfly stage when they are mirror images
        // -- the spectrum multiply and the x[0] step then happen in registers (lsm.h LSM_BFLY2).  A pre-multiplier that came in from a
        // six-step twiddle reaches x[0] through a one-word-per-instance stage of its own.
        std::vector<HostStage> second;
        {
            std::vector<HostStage> keep;
            keep.swap(prog->hstages);
            emit(I, TRN, fr, pd);
            second.swap(prog->hstages);
            prog->hstages.swap(keep);
        }
        if (failed || second.empty()) {
            failed = true;
            return;
        }
        HostStage& A = prog->hstages[last1];
        const HostStage& B = second.front();
        // (the forward stage may carry a pre-multiplier of its own when it is ALSO the first stage of its transform -- a one-stage leaf behind a
        // six-step twiddle, the shape of every prime whose p - 1 = m q with m <= 16: both tables are laid out by the slots of the same node, so
        // one index serves both up to a constant)
        const bool a_pre = (A.flags & LSM_PRE_MUL) != 0;
        const bool mirror = fuse && A.op == LSM_BFLY && B.op == LSM_BFLY && A.radix == B.radix && A.n == B.n && A.a == B.a && A.t == B.t && A.abase == B.abase &&
                            A.astep == B.astep && A.tw_kstep == B.tw_kstep && ((A.flags & LSM_TW_PRE) != 0) == ((B.flags & LSM_TW_POST) != 0) &&
                            (!(A.flags & LSM_TW_PRE) || A.tw_off == B.tw_off) && (B.flags & LSM_PRE_MUL) && !(B.flags & LSM_PRE_GLOBAL) &&
                            (!a_pre || (A.p == B.p && A.p_kstep == B.p_kstep));
        auto batch_digs = [&]() {
            std::vector<Dig> digs;
            for (size_t b = 0; b < fr.batch.size(); ++b) digs.push_back(Dig{fr.batch[b].n, fr.batch[b].stride, 0, pstride(pre, fr.batch[b].stride, (int)b < pre.n_outer), 0});
            return digs;
        };
        if (mirror) {
            HostStage fused = A;  // (by value: the stage list is about to change)
            prog->hstages.pop_back();
            // x[0] has not met an incoming pre-multiplier yet: the first inner transform's first stage took it for the other q - 1 inputs
            if (pre.on) push_stage(LSM_X0MUL, 1, 0, x0, 0, batch_digs(), 0, 0, pre, pre.tab_off + (x0 - pre.base) / pre.sigma, 0, 0.f);
            fused.op = LSM_BFLY2;
            fused.p2_delta = a_pre ? fused.p_off - B.p_off : 0;
            fused.flags = (fused.flags & LSM_TW_PRE) | LSM_PRE_MUL | (a_pre ? (LSM_PRE2 | (fused.flags & LSM_PRE_GLOBAL)) : 0);
            fused.p = B.p;
            fused.p_off = B.p_off;
            fused.p_kstep = B.p_kstep;
            fused.fix = true;
            fused.fix_off = x0 - s0;
            fused.fix_s0 = s0 - fr.base;
            fused.cfix = -(float)L;
            prog->hstages.push_back(fused);
            for (size_t i = 1; i < second.size(); ++i) prog->hstages.push_back(second[i]);
        } else {
            push_stage(LSM_FIX, 1, 0, s0, x0 - s0, batch_digs(), 0, 0, pre, pre.on ? pre.tab_off + (x0 - pre.base) / pre.sigma : 0, 0, -(float)L);
            for (auto& h : second) prog->hstages.push_back(h);
        }
    }
    void emit_mixed(const Node& nd, int mode, const Frame& fr, const PreOp& pre) {
        const Node &A = *nd.a, &B = *nd.b;
        const int PA = A.phys, N = nd.len;
        Frame fa = fr, fb = fr;
        fa.batch.push_back(Dim{B.phys, fr.sigma * PA});
        fb.sigma = fr.sigma * PA;
        fb.batch.push_back(Dim{A.phys, fr.sigma});
        // six-step twiddles by slot: row rho holds n2 = B.in^-1(rho), column c holds k1 = A.out^-1(c) (mixed_radix.rs:66-71)
        std::vector<int> binv(B.phys, -1), ainv(A.phys, -1);
        for (int i = 0; i < B.len; ++i) binv[B.in_pos[i]] = i;
        for (int i = 0; i < A.len; ++i) ainv[A.out_pos[i]] = i;
        int toff;
        auto it = mixed_tab.find(&nd);
        if (it != mixed_tab.end()) {
            toff = it->second;
        } else {
            std::vector<cd>& tab = tw_global ? prog->gtab : prog->ltab;
            toff = (int)tab.size();
            tab.resize(tab.size() + (size_t)nd.phys, cd(1, 0));
            for (int rho = 0; rho < B.phys; ++rho)
                for (int c = 0; c < PA; ++c)
                    if (binv[rho] >= 0 && ainv[c] >= 0) tab[toff + (size_t)rho * PA + c] = hooks->tw((size_t)binv[rho] * (size_t)ainv[c], (size_t)N);
            mixed_tab[&nd] = toff;
            ++n_mixed;
        }
        PreOp pt;
        pt.on = true;
        pt.global = tw_global;
        pt.tab_off = toff;
        pt.base = fr.base;
        pt.sigma = fr.sigma;
        pt.n_outer = (int)fr.batch.size();
        if (mode == FWD) {
            emit(A, FWD, fa, pre);
            emit(B, FWD, fb, pt);
        } else {
            emit(B, TRN, fb, pre);
            emit(A, TRN, fa, pt);
        }
    }
    void emit(const Node& nd, int mode, const Frame& fr, const PreOp& pre) {
        if (failed) return;
        if (nd.kind == LEAF)
            emit_leaf(nd, mode, fr, pre);
        else if (nd.kind == RADER)
            emit_rader(nd, fr, pre);
        else
            emit_mixed(nd, mode, fr, pre);
    }
};

// Lane order of a stage's work items.  The k-th access of a stage reads slot base + k astep in every lane, so two lanes of one LDS lane
// group collide exactly when their bases agree modulo the group's bank span: 32 eight-byte elements for the 32-lane groups of ds_read_b64,
// 16 sixteen-byte elements for the 16-lane groups of ds_read_b128 (MI355X_MICROARCH.md, LDS).  Work items are independent, so the host
// deals them into groups with distinct residues (largest residue classes first; what is left over goes where it fits) and sorts each group
// by residue, which also puts distinct residues modulo 16 into the 16-lane groups of the eight-byte writes.
inline void order_items(std::vector<unsigned>& dw, std::vector<unsigned>& pw, int group) {
    const size_t n = dw.size();
    if (n <= (size_t)group) return;
    std::vector<std::vector<size_t>> cls(group);
    for (size_t i = 0; i < n; ++i) cls[(dw[i] & 0xffffu) % (unsigned)group].push_back(i);
    std::vector<size_t> next(group, 0), order;
    order.reserve(n);
    size_t left = n;
    while (left > 0) {
        // one item of every class that still has items, the fullest classes first; a short group is topped up from the fullest classes
        std::vector<int> cl;
        for (int c = 0; c < group; ++c)
            if (next[c] < cls[c].size()) cl.push_back(c);
        std::sort(cl.begin(), cl.end(), [&](int x, int y) { return cls[x].size() - next[x] > cls[y].size() - next[y]; });
        std::vector<std::pair<int, size_t>> grp;
        for (int c : cl) grp.push_back({c, cls[c][next[c]++]});
        for (size_t fill = 0; grp.size() < (size_t)group && grp.size() < left; ++fill) {
            const int c = cl[fill % cl.size()];
            if (next[c] < cls[c].size()) grp.push_back({c, cls[c][next[c]++]});
            bool any = false;
            for (int d : cl) any = any || next[d] < cls[d].size();
            if (!any) break;
        }
        std::sort(grp.begin(), grp.end());
        for (auto& g : grp) order.push_back(g.second);
        left -= grp.size();
    }
    std::vector<unsigned> d2(n), p2(pw.size());
    for (size_t i = 0; i < n; ++i) {
        d2[i] = dw[order[i]];
        if (!pw.empty()) p2[i] = pw[order[i]];
    }
    dw.swap(d2);
    if (!pw.empty()) pw.swap(p2);
}

// expands the emitter's stages into the kernel's form for F rows per workgroup: stage headers + one descriptor word per work item of
// the workgroup, pre-multiplier indices behind them
inline bool finish(Program& prog, int esz) {
    const int F = prog.f;
    prog.stages.clear();
    prog.desc.clear();
    for (const HostStage& h : prog.hstages) {
        LsmStage st{};
        st.op = h.op;
        st.radix = h.radix;
        st.flags = h.flags;
        st.total = F * h.items;
        st.astep = h.astep;
        st.tw_kstep = h.tw_kstep;
        st.p_kstep = h.p_kstep;
        st.cfix = h.cfix;
        st.fix_off = h.fix_off;
        st.p2_delta = h.p2_delta;
        st.round = prog.nt * ((h.op == LSM_BFLY || h.op == LSM_BFLY2) ? items_per_thread(h.radix) : kLsmItems);
        std::vector<unsigned> dw, pd;
        for (int f = 0; f < F; ++f)
            for (int i = 0; i < h.items; ++i) {
                int rem = i, base = h.abase + f * prog.rp, tr = h.tw_off, pi = h.p_off;
                int rel = 0;  // the item's base relative to its Rader instance: S[0] sits at a known slot of the instance
                for (size_t d = 0; d < h.n.size(); ++d) {
                    const int dig = rem % h.n[d];
                    rem /= h.n[d];
                    base += dig * h.a[d];
                    tr += dig * h.t[d];
                    pi += dig * h.p[d];
                    if (h.inner[d] || h.p[d] != 0) rel += dig * h.a[d];  // (digits inside the instance: the transform's own and the batch digits its tables are indexed by)
                }
                if (h.fix && rel == h.fix_s0) pi |= (int)0x80000000u;
                if (base < 0 || base > 65535 || tr < 0 || tr > 65535) return false;
                dw.push_back((unsigned)base | ((unsigned)tr << 16));
                if (h.flags & LSM_PRE_MUL) pd.push_back((unsigned)pi);
            }
        if (prog.order_items) order_items(dw, pd, esz > 8 ? 16 : 32);
        st.desc_off = (int)prog.desc.size();
        prog.desc.insert(prog.desc.end(), dw.begin(), dw.end());
        if (h.flags & LSM_PRE_MUL) {
            st.pdesc_off = (int)prog.desc.size();
            prog.desc.insert(prog.desc.end(), pd.begin(), pd.end());
        }
        prog.stages.push_back(st);
    }
    return true;
}

inline bool build_program_nt(int n, int esz, int NT, const Hooks& hooks, Program& prog, size_t lds_budget, size_t lds_max, int force_f = 0) {
    prog = Program{};
#if defined(MI355_LSM_PLAIN_ORDER)
    prog.order_items = false;
#endif
    const int EMAX = kLsmEmax;
    Ctx cx;
    cx.nt = NT;
    cx.emax = EMAX;
    cx.row = n;
    if (n < 2 || (long long)n > (long long)NT * kLsmIoMax) return false;
    std::unique_ptr<Node> root = make_node(n, cx, 0);
    if (!root) return false;
    prog.n = n;
    prog.nt = NT;
    prog.rp = root->phys | 1;  // odd row pitch: lanes that walk across the rows of a workgroup fall on different banks
    // Two placements of the six-step tables: with the other stage tables in LDS (one fetch latency less per use) or in global memory (a
    // smaller workgroup: more rows or more workgroups per CU); the cost model picks.
    bool done = false;
    Program bestp;
    for (int pass = 0; pass < 2; ++pass) {
        prog.hstages.clear();
        prog.ltab.clear();
        prog.gtab.clear();
        Emitter em;
        em.prog = &prog;
        em.hooks = &hooks;
        em.tw_global = pass == 1;
#if defined(MI355_LSM_NO_FUSE)
        em.fuse = false;
#endif
        em.emit(*root, FWD, Frame{}, PreOp{});
        if (em.failed || prog.hstages.empty() || (int)prog.hstages.size() > kLsmMaxStages) return false;
        if (pass == 1 && em.n_mixed == 0) break;  // nothing to move
        // rows per workgroup: LOAD / STORE move at most kLsmIoMax elements per thread; 16-bit slots; LDS
        int fmax = std::min((NT * kLsmIoMax) / n, 65535 / prog.rp);
        if (fmax < 1) return false;
        const size_t tab = prog.ltab.size();
        if (tab > 60000) continue;  // a twiddle row is a 16-bit field of the descriptor word
        auto lds_for = [&](int f) { return ((size_t)f * prog.rp + tab) * (size_t)esz; };
        while (fmax > 1 && lds_for(fmax) > lds_budget) --fmax;
        if (lds_for(fmax) > lds_max) continue;
        // rows per workgroup by a cost model: SIMD time per row = sum over stages of (rounds x item cost + a fixed barrier / latency term) x
        // waves, over F rows -- a stage with F items = 1.1 NT pays two rounds for the work of one, a small F pays the barriers alone
        int F = fmax;
        double best_t = 1e30;
        std::vector<double> cost_of(fmax + 1, 0.0);
        for (int f = 1; f <= fmax; ++f) {
            double t = 0;
            for (auto& st : prog.hstages) {
                const int rounds = (f * st.items + NT - 1) / NT;
                const int ipt = (st.op == LSM_BFLY || st.op == LSM_BFLY2) ? items_per_thread(st.radix) : kLsmItems;
                if (rounds > ipt) t += 150.0 * ((rounds + ipt - 1) / ipt - 1);  // later rounds fetch their descriptor words in line
                if (st.flags & LSM_PRE_GLOBAL) t += 60.0;                       // a table in global memory: its latency is not hidden by the stage
                t += rounds * (st.op == LSM_BFLY ? item_cost(st.radix) : st.op == LSM_BFLY2 ? 1.8 * item_cost(st.radix) : 20.0) + 40.0 + 10.0 * (NT / 64.0);  // (a barrier costs more the more waves meet at it)
            }
            t += 2.0 * (((double)f * n + NT - 1) / NT) * 8.0 + 200.0;  // LOAD / STORE
            t = t * (NT / 64.0) / f;
            // resident waves hide the stages' LDS round trips and barriers: what the workgroup's LDS lets a CU hold (160 KiB; at most 16 waves
            // at 128 VGPRs), counted against the dozen that keeps the SIMDs busy
            const double wgs = std::max(1.0, std::min(32.0, std::floor(160.0 * 1024 / (double)lds_for(f))));
            const double waves = std::min(16.0, wgs * (NT / 64.0));
            t *= 12.0 / std::min(12.0, waves);
            if (force_f > 0 && f == std::min(force_f, fmax)) {
                best_t = t;
                F = f;
            }
            if (force_f == 0) cost_of[f] = t;
        }
        if (force_f == 0) {
            // Rows per workgroup from the MEASURED optimum (profiles/r6/lsm_ntf_sweep_f32.jsonl: 16 lengths x every block size x up to ten row
            // counts): the rate peaks where a thread holds 13 .. 24 elements of the workgroup's rows per stage -- enough work between two
            // barriers to hide them, within the four items a thread may hold -- and is flat to 3 % across block sizes at that load, so: the
            // row count that brings the workgroup closest to 16 elements per thread; the cost above only ranks the block sizes
            const double want = 16.0;
            F = std::max(1, std::min(fmax, (int)std::lround(want * NT / n)));
            const double e = (double)F * n / NT;
            best_t = cost_of[F] * (e < 12.0 ? 1.0 + 0.12 * (12.0 - e) : e > 26.0 ? 1.0 + 0.04 * (e - 26.0) : 1.0) * (1.0 + 0.0002 * NT);
        }
        if (!done || best_t < bestp.est) {
            prog.est = best_t;
            prog.f = F;
            prog.tab_off = F * prog.rp;
            prog.lds_elems = (size_t)F * prog.rp + tab;
            bestp = prog;
            done = true;
        }
    }
    if (done) prog = bestp;
    if (!done || !finish(prog, esz)) return false;
    prog.ldperm.resize((size_t)prog.f * n);
    prog.stperm.resize((size_t)prog.f * n);
    for (int f = 0; f < prog.f; ++f)
        for (int i = 0; i < n; ++i) {
            prog.ldperm[(size_t)f * n + i] = (unsigned short)(f * prog.rp + root->in_pos[i]);
            prog.stperm[(size_t)f * n + i] = (unsigned short)(f * prog.rp + root->out_pos[i]);
        }
    prog.desc_str = describe(*root);
    prog.root_kind = root->kind;
    return true;
}
// Builds the program for length n (elements of `esz` bytes): workgroups of 64 .. 1024 threads, the one the cost model likes best.  Returns false when the length has no tree within the machine's limits (registers, LDS, 16-bit slots).
// (force_nt / force_f: measurement builds pin the block size and the rows per workgroup)
inline bool build_program(int n, int esz, const Hooks& hooks, Program& prog, size_t lds_budget = 64 * 1024, size_t lds_max = 160 * 1024, int force_nt = 0, int force_f = 0) {
    bool have = false;
    for (int nt : {64, 128, 256, 512, 1024}) {
        if (force_nt > 0 && nt != force_nt) continue;
        if (nt == 1024 && esz > 8) break;  // Complex<f64>: the 1024-thread kernel would have to live in 128 VGPRs (it spills)
        Program cand;
        if (!build_program_nt(n, esz, nt, hooks, cand, lds_budget, lds_max, force_f)) continue;
        if (!have || cand.est < prog.est) prog = std::move(cand);
        have = true;
    }
    return have;
}

}  // namespace lsm
}  // namespace mi355

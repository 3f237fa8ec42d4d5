// sg_layout.cpp -- tile-padded device layouts of the policy / discriminator parameter vectors and
// the conversion to and from the reference's flat state_dict order
// (Policy: a2c/model.py:37-114,233-264 + a2c/distributions.py:91-118; SplitPolicy:
// a2c/model_split.py:39-95,157-238; Discriminator: a2c/algo/gail.py:40-43).
#include <string.h>

#include <functional>

#include "sg_common.h"

static SgTrunk make_trunk(int& cursor, int H, int ldO, int P, int EX) {
    SgTrunk t;
    const int Hp = SG_PAD16(H), ldH = SG_LD(H);
    t.H = H; t.Hp = Hp; t.ldH = ldH;
    t.off = cursor;
    int o = 0;
    t.w1 = o; o += Hp * ldO;
    t.b1 = o; o += Hp;
    t.w2 = o; o += Hp * ldH;
    t.b2 = o; o += Hp;
    t.P = P;
    t.Pp = SG_PAD16(P);
    t.ldP = SG_LD(P);
    t.wh = o; o += t.Pp * ldH;
    t.bh = o; o += t.Pp;
    t.EX = EX;
    t.ex = o; o += EX ? SG_PAD16(EX) : 0;
    t.size = o;
    cursor += o;
    return t;
}

SgPolicyDesc sg_make_policy_desc(int kind, int O, int A, int H, int num_feet, int Hc) {
    SgPolicyDesc d;
    memset(&d, 0, sizeof d);
    if (Hc <= 0) Hc = H;
    d.kind = kind; d.O = O; d.A = A; d.H = H; d.num_feet = num_feet; d.Hc = Hc;
    d.Op = SG_PAD16(O); d.ldO = SG_LD(O);
    const int Hmax = H > Hc ? H : Hc;
    d.Hp = SG_PAD16(Hmax); d.ldH = SG_LD(Hmax);
    int cur = 0;
    if (kind == SG_POLICY_MLP) {
        d.n_trunks = 2;
        d.trunk[0] = make_trunk(cur, H, d.ldO, A, A);    // actor: fc_mean head + logstd
        d.trunk[1] = make_trunk(cur, Hc, d.ldO, 1, 0);   // critic: critic_linear head
    } else {
        d.n_trunks = 3;
        d.nc = 4 * num_feet; d.na = 3 * num_feet;
        d.trunk[0] = make_trunk(cur, H, d.ldO, 2 * d.nc, 0);  // contact: [mean | logstd]
        d.trunk[1] = make_trunk(cur, H, d.ldO, 2 * d.na, 0);  // actuator: [mean | logstd]
        d.trunk[2] = make_trunk(cur, Hc, d.ldO, 1, 0);        // critic_full
    }
    d.total = cur;
    return d;
}

// Enumerates the dense segments of the flat vector in state_dict order:
// f(flat_off, padded_off, rows, cols, padded_ld)
typedef std::function<void(int64_t, int64_t, int, int, int)> SegFn;

static int64_t policy_segments(const SgPolicyDesc& d, const SegFn& f) {
    int64_t fo = 0;
    const int O = d.O, H = d.H, A = d.A;
    auto mat = [&](int64_t po, int rows, int cols, int ld) { f(fo, po, rows, cols, ld); fo += (int64_t)rows * cols; };
    for (int t = 0; t < d.n_trunks; ++t) {
        const SgTrunk& tr = d.trunk[t];
        mat(tr.off + tr.w1, tr.H, O, d.ldO);
        mat(tr.off + tr.b1, 1, tr.H, tr.H);
        mat(tr.off + tr.w2, tr.H, tr.H, tr.ldH);
        mat(tr.off + tr.b2, 1, tr.H, tr.H);
    }
    const SgTrunk& cr = d.trunk[d.n_trunks - 1];
    mat(cr.off + cr.wh, 1, cr.H, cr.ldH);   // critic_linear / critic_full.4 weight [1, critic width]
    mat(cr.off + cr.bh, 1, 1, 1);
    if (d.kind == SG_POLICY_MLP) {
        const SgTrunk& ac = d.trunk[0];
        mat(ac.off + ac.wh, A, H, ac.ldH);   // fc_mean
        mat(ac.off + ac.bh, 1, A, A);
        mat(ac.off + ac.ex, 1, A, A);       // logstd._bias [A,1]
    } else {
        const SgTrunk& c = d.trunk[0];
        const SgTrunk& a = d.trunk[1];
        mat(c.off + c.wh, d.nc, H, c.ldH);                 // contact_mean
        mat(c.off + c.bh, 1, d.nc, d.nc);
        mat(a.off + a.wh, d.na, H, a.ldH);                 // actuator_mean
        mat(a.off + a.bh, 1, d.na, d.na);
        mat(c.off + c.wh + d.nc * c.ldH, d.nc, H, c.ldH);  // contact_logstd -> head rows [nc, 2nc)
        mat(c.off + c.bh + d.nc, 1, d.nc, d.nc);
        mat(a.off + a.wh + d.na * a.ldH, d.na, H, a.ldH);  // actuator_logstd -> head rows [na, 2na)
        mat(a.off + a.bh + d.na, 1, d.na, d.na);
    }
    return fo;
}

int64_t sg_policy_flat_count(const SgPolicyDesc& d) {
    return policy_segments(d, [](int64_t, int64_t, int, int, int) {});
}

void sg_policy_pad(const SgPolicyDesc& d, const float* flat, float* padded) {
    policy_segments(d, [&](int64_t fo, int64_t po, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(padded + po + (int64_t)r * ld, flat + fo + (int64_t)r * cols, sizeof(float) * cols);
    });
}

void sg_policy_unpad(const SgPolicyDesc& d, const float* padded, float* flat) {
    policy_segments(d, [&](int64_t fo, int64_t po, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(flat + fo + (int64_t)r * cols, padded + po + (int64_t)r * ld, sizeof(float) * cols);
    });
}

SgDiscDesc sg_make_disc_desc(int F, int Hd) {
    SgDiscDesc d;
    d.F = F; d.Hd = Hd;
    d.Fp = SG_PAD16(F); d.ldF = SG_LD(F);
    d.Hp = SG_PAD16(Hd); d.ldH = SG_LD(Hd);
    int o = 0;
    d.w1 = o; o += d.Hp * d.ldF;
    d.b1 = o; o += d.Hp;
    d.w2 = o; o += d.Hp * d.ldH;
    d.b2 = o; o += d.Hp;
    d.w3 = o; o += d.Hp;
    d.b3 = o; o += 16;
    d.total = o;
    return d;
}

static int64_t disc_segments(const SgDiscDesc& d, const SegFn& f) {
    int64_t fo = 0;
    auto mat = [&](int64_t po, int rows, int cols, int ld) { f(fo, po, rows, cols, ld); fo += (int64_t)rows * cols; };
    mat(d.w1, d.Hd, d.F, d.ldF);
    mat(d.b1, 1, d.Hd, d.Hd);
    mat(d.w2, d.Hd, d.Hd, d.ldH);
    mat(d.b2, 1, d.Hd, d.Hd);
    mat(d.w3, 1, d.Hd, d.Hd);
    mat(d.b3, 1, 1, 1);
    return fo;
}

int64_t sg_disc_flat_count(const SgDiscDesc& d) {
    return disc_segments(d, [](int64_t, int64_t, int, int, int) {});
}

void sg_disc_pad(const SgDiscDesc& d, const float* flat, float* padded) {
    disc_segments(d, [&](int64_t fo, int64_t po, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(padded + po + (int64_t)r * ld, flat + fo + (int64_t)r * cols, sizeof(float) * cols);
    });
}

void sg_disc_unpad(const SgDiscDesc& d, const float* padded, float* flat) {
    disc_segments(d, [&](int64_t fo, int64_t po, int rows, int cols, int ld) {
        for (int r = 0; r < rows; ++r) memcpy(flat + fo + (int64_t)r * cols, padded + po + (int64_t)r * ld, sizeof(float) * cols);
    });
}

// cond_dims.hpp — sizes and scratch layout of the partially condensed QP of one instance (cond_ipm.hpp), host and device.
#pragma once
#include "params.hpp"
#include <algorithm>

namespace usv {

// sizes and scratch offsets (doubles) of the condensed QP of one instance; host and device
struct CondDims {
    int Mb, N2, nuh, nzh, nxr, R, nrows, nbu, nbx, K;   // Mb: stages of the LONGEST block (sizes everything)
    int N1, R1;           // HPIPM's partition: N1 = N / N2 stages per block, the first R1 = N - N2 N1 blocks one more
    int xr[LANES];        // states that some row touches (bounded, position), ascending
    int xr_of[LANES];     // state -> index in xr, or -1
    int uvar[LANES];      // u rows: control index;  xvar: x rows: index into xr
    int xvar[LANES];
    int ipx, ipy;         // index in xr of the position states (-1 without obstacle rows)
    // per block
    long o_SR, o_cr, o_BA, o_bt, o_H0, o_g0, o_row, o_Luu, o_P, o_Pb, o_w, o_pi, o_rg, o_rb, o_dwa, o_dw, o_dpi, o_p, o_lus, o_dg, blk;
    long o_cdel, o_cdela, o_cdelf; // the touched states' values at the iterate / along the affine and the final step, kept from the sweep that computes them
    long total;           // (N2 + 1) * blk
    long lds_doubles;     // LDS the kernel needs (doubles), for NT threads
    int nt;               // threads of a team (cond_prepare picks it: the instantiation with the most resident waves per CU)
};

inline bool cond_dims(const DevSpec &S, int nx, int nu, int ipx, int ipy, int N2, int nt, bool soft, CondDims &D)
{
    if (N2 < 1 || N2 >= S.N) return false;
    D.N1 = S.N / N2; D.R1 = S.N - N2 * D.N1;
    D.Mb = D.N1 + (D.R1 > 0 ? 1 : 0); D.N2 = N2; D.nuh = D.Mb * nu; D.nzh = D.nuh + nx; D.K = S.K;
    if (D.nzh > 64) return false; // (one wave holds a block vector in the triangular solves; the index table packs rows in 8 bits)
    D.nbu = D.nbx = D.nxr = 0;
    for (int i = 0; i < LANES; i++) { D.xr_of[i] = -1; D.xr[i] = D.uvar[i] = D.xvar[i] = 0; }
    for (int l = 0; l < nu; l++) if (S.has_b[l]) D.uvar[D.nbu++] = l;
    for (int s = 0; s < nx; s++)
        if (S.has_b[nu + s] || (S.K > 0 && (s == ipx || s == ipy))) { D.xr_of[s] = D.nxr; D.xr[D.nxr++] = s; }
    for (int s = 0; s < nx; s++) if (S.has_b[nu + s]) D.xvar[D.nbx++] = D.xr_of[s];
    D.ipx = S.K > 0 ? D.xr_of[ipx] : -1; D.ipy = S.K > 0 ? D.xr_of[ipy] : -1;
    D.R = D.nbu + D.nbx + S.K; D.nrows = D.Mb * D.R;
    long o = 0;
    auto take = [&](long n) { const long at = o; o += (n + 15) / 16 * 16; return at; }; // 128-byte pieces
    const int nz = nx + nu;
    // the matrix group: what every sweep brings into LDS for a block, in ONE piece - the kernel's LDS holds the same pieces in the same order
    // at the same distances (CondIpm's constructor), so that the block comes in as one linear copy with all of a thread's loads in flight
    D.o_SR = take(std::max<long>((long)D.Mb * D.nxr * D.nzh, 2L * nx * D.nzh + (long)nz * D.nzh)); // (the LDS piece doubles as (Sm, Sn, Tm) while condensing)
    D.o_cr = take((long)D.Mb * D.nxr);
    D.o_BA = take((long)nx * D.nzh); D.o_bt = take(nx);
    D.o_g0 = take(D.nzh); D.o_H0 = take((long)D.nzh * D.nzh);
    D.o_row = take((soft ? 14L : 8L) * D.nrows); // ll, lu, tl, tu, dl, du, cx, cy (+ sl, su, lsl, lsu, tsl, tsu when the obstacle rows are soft)
    D.o_Luu = take((long)D.nzh * D.nzh);     // the eliminated stage matrix as it stands in LDS ([Luu; Lxu] in its first nuh columns)
    D.o_P = take((long)nx * nx);             // P_{i+1}
    D.o_Pb = take(nx);
    D.o_w = take(D.nzh); D.o_pi = take(nx); D.o_rg = take(D.nzh); D.o_rb = take(nx);
    D.o_dwa = take(D.nzh); D.o_dw = take(D.nzh); D.o_dpi = take(nx); D.o_p = take(nx); D.o_lus = take(D.nuh); D.o_dg = take(D.nuh);
    D.o_cdel = take((long)D.Mb * D.nxr); D.o_cdela = take((long)D.Mb * D.nxr); D.o_cdelf = take((long)D.Mb * D.nxr);
    D.blk = o;
    D.total = (long)(N2 + 1) * D.blk;
    long l = 0;
    l += (D.o_H0 - D.o_SR) + (long)(D.nzh + 1) * D.nzh + 16;              // the matrix group: SRm | vcr | BAm | vbt | vg0 | Gm
    l += (long)nx * std::max(D.nzh, nz);                                  // PBm | BAk
    l += (long)nx * nx;                                                   // Pn
    l += 4L * D.Mb * D.nxr + 3L * D.Mb * D.nxr + D.Mb + 3L * D.nuh;       // expansions, slots
    l += 8;                                                               // reductions
    l += 2L * LANES + (D.nzh * (D.nzh + 1) / 2 + 3) / 4 + 3L * LANES + (soft ? 7L : 1L) * KMAX; // short tables (ints), triangle index table (16 bits), spec copies
    l += 7L * D.nzh + 10L * nx + 2L * D.nuh + 2L * nz + (long)D.Mb * nz; // vectors
    D.lds_doubles = l + 8;   // (LDS is handed out in pieces of 1280 bytes, 128 to a CU: the 30-variable blocks of usv_model_pf_ca take 32 - four teams per CU)
    D.nt = nt;
    return true;
}

} // namespace usv

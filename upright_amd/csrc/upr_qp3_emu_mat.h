// upr_qp3_emu_mat.h -- the HOST EMULATION's matrix sweep of upr_qp3.h: the body of upr_qp3<C>::backward_mat() under UPR_HOST_EMU
// (tests/emu/upr_emu.cpp; included inside that member function, nowhere else).  It is the four-wave form of rounds 1 - 2 --
// three workgroup barriers and five LDS round trips a knot, plain loops, no cross-lane hardware -- whose device branches were
// deleted in round 6: the device runs backward_mat_sw2().  Same arithmetic knot by knot, so the CPU tests of the kernel source
// (tests/test_emu.py) check the recursion the device implements with two cooperating waves.
        // The four-wave form of rounds 1 - 2 (three workgroup barriers and five LDS round trips a knot), kept as the HOST EMULATION's
        // matrix sweep (tests/emu): plain loops, no cross-lane hardware.  Its device branches were deleted in round 6.
        {
        const double irho = 1.0 / UPR_QP_RHO_N;
        double* Pc = L + O::Pa; double* Pn = L + O::Pb;
        {
            const double* Ck = rec(N - 1) + lin_gx;
            UPR_FORT(e, NE * NX) L[O::ck + e] = Ck[e];
            UPR_FORT(e, C::NLS) L[O::lsik + e] = G[F::lsi + (N - 1) * C::NLS + e];
            UPR_FORT(e, C::NH) L[O::heek + ((N - 1) & 1) * O::r2(C::NH) + e] = G[hee_w + (N - 1) * C::NH + e];
        }
        UPR_FORT(e, NX * NX) {
            const int i = e / NX, j = e % NX;
            double v = (i == j) ? LK[O::wx + N * NX + i] : 0.0;
            if (neN > 0) {
                if (i < NQ && j < NQ) { for (int q = 0; q < 3; ++q) v += irho * L[O::jN + q * NQ + i] * L[O::jN + q * NQ + j]; }
                else if (i == j) v += irho;
            }
            Pc[e] = v;
        }
        UPR_SYNC();
        // Vc = Lsi C by jobs of three rows (rows 0-2 or 3-5 of a column): 2 nx jobs, which fit ONE wave.  A wave that holds two
        // job kinds runs them one after the other; with row pairs (3 nx jobs) the Vc jobs spilled onto the wave that also
        // carries the overflow of the A'P+A jobs, and phase 1 waited for that wave (profiles/r01h_mat_waves.txt).
        static_assert(NE % 3 == 0, "three rows per Vc job");
        constexpr int NVC = (NE / 3) * NX;
        constexpr int VC0 = (NT >= 256) ? (C::NB > 1 ? 96 : 128) : NQ * NQ;   // first lane of the Vc jobs: a wave of their own where there is one (a wave and a half for the multi-body shapes)
        for (int k = N - 1; k >= 0; --k) {
            // next knot's C, Lsi, Hee: global -> registers now, -> LDS after the barrier (their readers are in phase 1 /
            // in the accumulator preload of the NEXT knot).  (Multi-body shapes: requesting them behind the barrier of phase 1
            // instead takes ~14 k cycles per iteration out of phase 1 -- every reload of a spilled register there waits for
            // all memory operations in flight -- and puts them back into phases 2 and 3: measured no different.)
            constexpr int NLS = C::NLS, NPF = NE * NX + NLS + C::NH, CKQ = (NPF + NT - 1) / NT;
            double ckn[CKQ];
            const int tid_ = tid();
#pragma unroll
            for (int q = 0; q < CKQ; ++q) {
                const int f = tid_ + q * NT;
                double v = 0.0;
                if (k > 0) {
                    if (f < NE * NX) v = rec(k - 1)[lin_gx + f];
                    else if (f < NE * NX + NLS) v = G[F::lsi + (k - 1) * NLS + (f - NE * NX)];
                    else if (f < NPF) v = G[hee_w + (k - 1) * C::NH + (f - NE * NX - NLS)];
                }
                ckn[q] = v;
            }
            // phase 1.  Jobs [0, NQ*NQ): lane (ii, jj) loads the 9 block entries P+[(a,ii)][(c,jj)] once and emits the
            // entries of A'P+A it owns (upper triangle of the result), 3 of Hux = B'P+A and 1 of Hjj = B'P+B + R + W;
            // then NVC jobs of Vc = Lsi C and NX jobs of P+ b.
            // job -> lane map: A'P+A blocks from lane 0, Vc row pairs behind them, P+ b on the LAST wave of the workgroup
            // (a wave that holds two job kinds runs them one after the other: the 27-term dot products must not share
            // a wave with anything else)
            constexpr int PB0 = (NT >= 256) ? NT - 64 : ((NQ * NQ + NVC + 1) & ~1);
            static_assert(VC0 >= NQ * NQ && PB0 > VC0 && (C::NB > 1 || PB0 >= VC0 + NVC) && PB0 % 2 == 0, "jobs overlap / P+ b lane pairs start on an even lane");
            constexpr int NPB = NX;
            constexpr bool VC_MFMA = false;
            UPR_FORT(e, PB0 + NPB) {
                if (e < NQ * NQ) {
                    const int ii = e / NQ, jj = e % NQ;
                    double p[3][3];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) p[a][c] = Pc[(a * NQ + ii) * NX + c * NQ + jj];
                    // T = P A (columns), then A' T (rows); A = [[1,h,h2],[0,1,h],[0,0,1]]
                    double t[3][3], o2[3][3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) { t[a][0] = p[a][0]; t[a][1] = h * p[a][0] + p[a][1]; t[a][2] = h2 * p[a][0] + h * p[a][1] + p[a][2]; }
                    if (k > 0) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) { o2[0][c] = t[0][c]; o2[1][c] = h * t[0][c] + t[1][c]; o2[2][c] = h2 * t[0][c] + h * t[1][c] + t[2][c]; }
                        // only the upper triangle of A'P+A is read back (and mirrored): rounding cannot make P asymmetric
#pragma unroll
                        for (int a = 0; a < 3; ++a)
#pragma unroll
                            for (int c = 0; c < 3; ++c) if (a < c || (a == c && ii <= jj)) Pn[(a * NQ + ii) * NX + c * NQ + jj] = o2[a][c];
                    }
                    // B' (P A) = h3 t[0] + h2 t[1] + h t[2]   (knot 0: only for the feedback gain K_0)
                    if (k > 0 || fbk) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) L[O::hux + ii * NX + c * NQ + jj] = h3 * t[0][c] + h2 * t[1][c] + h * t[2][c];
                    }
                    // B' P B
                    double v = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) v += coefB(a) * (h3 * p[a][0] + h2 * p[a][1] + h * p[a][2]);
                    if (ii == jj) v += h * L[O::rd + ii] + LK[O::wu + k * NU + ii];
                    // lower triangle, packed: the factoring wave reads it with paired 128-bit loads (it is the wave phase 2 waits for)
                    if (jj <= ii) L[O::hjj + ii * (ii + 1) / 2 + jj] = v;
                } else if (e >= VC0 && e < ((NVC <= PB0 - VC0) ? VC0 + NVC : PB0)) {
                    // (the multi-body shapes have more jobs than lanes between VC0 and PB0: those lanes take several)
                    if (C::MULTI) {
                        // star arrangements: a job is one column of one body's block, all six rows (static triangular loops)
                        if (k > 0) for (int f = e - VC0; f < C::NB * NX; f += PB0 - VC0) {
                            const int blk = f / NX, c = f % NX, bo = 6 * blk;
                            const double* Ls = L + O::lsik + 36 * blk;
                            double cm[6];
#pragma unroll
                            for (int m = 0; m < 6; ++m) cm[m] = L[O::ck + (bo + m) * NX + c];
#pragma unroll
                            for (int r = 0; r < 6; ++r) { double v = 0.0;
#pragma unroll
                                for (int m = 0; m <= r; ++m) v += Ls[r * 6 + m] * cm[m];
                                L[O::vc + (bo + r) * NX + c] = v; }
                        }
                    } else
                    if (!VC_MFMA && k > 0) for (int f = e - VC0; f < NVC; f += PB0 - VC0) {
                        // three rows (r0 .. r0 + 2 of the knot) of the block of body g / 2: Vc = blockdiag(Lsi_b) C
                        constexpr int SBV = C::SB;
                        const int g = f / NX, c = f % NX, r0 = 3 * g, blk = r0 / SBV, bo = SBV * blk, q0 = r0 - bo;
                        const double* Ls = L + O::lsik + SBV * SBV * blk;
                        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                        // full-length rows with the entries above the diagonal masked: no lane-dependent trip count
#pragma unroll
                        for (int m = 0; m < SBV; ++m) {
                            const double cm = L[O::ck + (bo + m) * NX + c];
                            const double l0 = Ls[q0 * SBV + m], l1 = Ls[(q0 + 1) * SBV + m], l2 = Ls[(q0 + 2) * SBV + m];
                            v0 += ((m <= q0) ? l0 : 0.0) * cm; v1 += ((m <= q0 + 1) ? l1 : 0.0) * cm; v2 += ((m <= q0 + 2) ? l2 : 0.0) * cm;
                        }
                        L[O::vc + r0 * NX + c] = v0; L[O::vc + (r0 + 1) * NX + c] = v1; L[O::vc + (r0 + 2) * NX + c] = v2;
                    }
                } else if (e >= PB0) {
                    const int i = e - PB0;
                    double p0 = 0.0, p1 = 0.0, p2 = 0.0;   // three independent chains
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        p0 += Pc[i * NX + j] * LK[O::bks + k * NX + j];
                        p1 += Pc[i * NX + NQ + j] * LK[O::bks + k * NX + NQ + j];
                        p2 += Pc[i * NX + 2 * NQ + j] * LK[O::bks + k * NX + 2 * NQ + j];
                    }
                    LK[O::Pbs + k * NX + i] = (p0 + p1) + p2;
                }
            }
            mtoc(0);
            sync_lds();
            toc(6);
            if (k + 1 < N) UPR_FORT(c, NX) feedback_column(k + 1, c);
            // wave 0: EVERY lane factors Hjj for itself in registers (right-looking, 9 dependent pivots) and carries its own
            // column of Hux through the same eliminations: V = Lj^-1 Hux comes out with the factor and nothing is
            // exchanged between lanes.  (The one-column-per-lane form of the same elimination broadcast each multiplier
            // through a v_readlane pair: 94 of them per knot at ~20 cycles of latency each, 4.4 k cycles per knot.)
            if (wave0()) {
                UPR_SETPRIO(3);
                if (tid() == 0) {
                    constexpr int NM = NQ + NX;
                    double M[NQ][NM];
                    for (int j = 0; j < NQ; ++j) for (int c = 0; c < NM; ++c) M[j][c] = (c < NQ) ? L[O::hjj + (j >= c ? j * (j + 1) / 2 + c : c * (c + 1) / 2 + j)] : ((k > 0 || fbk) ? L[O::hux + j * NX + (c - NQ)] : 0.0);
                    for (int p2 = 0; p2 < NQ; ++p2) {
                        double piv = M[p2][p2];
                        if (!(piv > 0.0)) { L[O::misc] = 1.0; piv = 1.0; }
                        const double idg = upr_rsqrt(piv);
                        double y[NM];
                        for (int c = 0; c < NM; ++c) y[c] = M[p2][c] * idg;
                        for (int j = p2 + 1; j < NQ; ++j) for (int c = 0; c < NM; ++c) M[j][c] -= y[c] * y[j];
                        for (int c = 0; c < NM; ++c) M[p2][c] = (c == p2) ? idg : y[c];
                    }
                    for (int c = 0; c < NQ; ++c) for (int p2 = 0; p2 <= c; ++p2) { G[F::Ljis + k * C::NH + c * (c + 1) / 2 + p2] = M[p2][c]; L[lkb(k) + c * (c + 1) / 2 + p2] = M[p2][c]; }
                    if (k > 0 || fbk) for (int c = 0; c < NX; ++c) for (int p2 = 0; p2 < NQ; ++p2) L[vmb(k) + p2 * NX + c] = M[p2][NQ + c];
                }
            }
                if (k == 0) break;
            mtoc(2);
            sync_lds();
            toc(8);
            // P = sym(A'P+A) + Q~ + Vc'Vc - V'V (upper triangle, mirrored)
            UPR_FORT(e, NX * NX) {
                const int i = e / NX, j = e % NX;
                if (i <= j) {
                    double v = Pn[i * NX + j];
                    if (i == j) v += h * L[O::qd + i] + LK[O::wx + k * NX + i];
                    if (j < NQ) v += h * L[O::heek + (k & 1) * O::r2(C::NH) + upr_tri(NQ, i, j)];
                    for (int q = 0; q < NE; ++q) v += L[O::vc + q * NX + i] * L[O::vc + q * NX + j];
                    for (int m = 0; m < NQ; ++m) v -= L[vmb(k) + m * NX + i] * L[vmb(k) + m * NX + j];
                    Pc[i * NX + j] = v; Pc[j * NX + i] = v;
                }
            }
            {
#pragma unroll
                for (int q = 0; q < CKQ; ++q) {
                    const int f = tid_ + q * NT;
                    if (f < NE * NX) L[O::ck + f] = ckn[q];
                    else if (f < NE * NX + NLS) L[O::lsik + (f - NE * NX)] = ckn[q];
                    else if (f < NPF) L[O::heek + ((k - 1) & 1) * O::r2(C::NH) + (f - NE * NX - NLS)] = ckn[q];
                }
            }
            mtoc(4);
            sync_lds();
            toc(9);
        }
        UPR_SETPRIO(0);
        UPR_SYNC();
        // knot 0 has no successor in the loop: its feedback (wanted only for the linear policy) is formed here
        if (fbk) UPR_FORT(c, NX) feedback_column(0, c);
        UPR_SYNC();
        }

// The Groth16 prover around the multi-scalar sums, generic over the curve; included inside the scalar field's
// namespace after frntt_impl.cuh with
//   G16_CW                      u32 words per base-field element (8: BN254, 12: BLS12-381)
//   G16_MSM_G1 / G16_MSM_G2     the curve's multi-scalar sums (msm.hip)
//   G16_FQ_TO_MONT / _TO_CANON  base-field conversions in place (msm.hip)
//   G16_JSON_CURVE              "BN128" / "BLS12381" (json_utils.rs:305-315)
//   G16_FN(name)                exported factory name
// Restates bellman_ce groth16/prover.rs create_proof with explicit r, s (the reference draws them from its rng,
// groth16/src/groth16.rs:88-96) over the circuit algebraic/src/circom_circuit.rs:94-160 synthesises.
// No include guard on purpose.

struct G16_FN(SetupImpl) final : Groth16Setup {
    uint32_t ni = 0, n_aux = 0, n_wires = 0;
    int logm = 0;
    u64 m = 0, n_rows = 0;
    DevBuf rp[3], cl[3], cf[3];
    DevBuf a_idx, b_idx, l_idx;
    u64 na = 0, nb = 0, nh = 0;
    DevBuf h, l, a, b1, b2;
    std::vector<u32> alpha1, beta1, delta1, beta2, delta2;   // Montgomery, host copies for the final sums
    std::vector<std::vector<u32>> ic;

    static constexpr size_t P1 = 2 * G16_CW, P2 = 4 * G16_CW;   // u32 words per affine point

    G16_FN(SetupImpl)(const g16::R1cs& rc, const g16::Params& pk) {
        hipStream_t st = nullptr;
        ni = 1 + rc.n_pub_out + rc.n_pub_in;
        n_wires = rc.n_wires;
        ZK_REQUIRE(n_wires >= ni, "groth16: r1cs header: fewer wires than public signals");
        n_aux = n_wires - ni;
        // circom_circuit.rs:143-157: rows with (A or B empty) and C empty are not enforced; prover.rs then appends
        // one `input_i * 0 = 0` row per input
        std::vector<const g16::Row*> rows;
        for (const auto& r : rc.rows)
            if (!((r.lc[0].col.empty() || r.lc[1].col.empty()) && r.lc[2].col.empty())) rows.push_back(&r);
        n_rows = rows.size() + ni;
        logm = 0;
        while ((1ull << logm) < n_rows) ++logm;
        m = 1ull << logm;
        std::vector<char> a_aux(n_wires, 0), b_any(n_wires, 0);
        for (int w = 0; w < 3; ++w) {
            std::vector<u64> ptr; std::vector<u32> cols, coef;
            ptr.push_back(0);
            for (const g16::Row* r : rows) {
                const auto& lc = r->lc[w];
                for (size_t k = 0; k < lc.col.size(); ++k) {
                    ZK_REQUIRE(lc.col[k] < n_wires, "groth16: r1cs: wire index out of range");
                    cols.push_back(lc.col[k]);
                    coef.insert(coef.end(), lc.coeff.begin() + 8 * k, lc.coeff.begin() + 8 * k + 8);
                    if (w == 0 && lc.col[k] >= ni) a_aux[lc.col[k]] = 1;
                    if (w == 1) b_any[lc.col[k]] = 1;
                }
                ptr.push_back(cols.size());
            }
            for (uint32_t i = 0; i < ni; ++i) {
                if (w == 0) { cols.push_back(i); const u32 one[8] = {1, 0, 0, 0, 0, 0, 0, 0}; coef.insert(coef.end(), one, one + 8); }
                ptr.push_back(cols.size());
            }
            rp[w].reserve(ptr.size() * 8); cl[w].reserve(cols.size() * 4 + 4); cf[w].reserve(cols.size() * NR * 4 + 4);
            ZK_HIP(hipMemcpy(rp[w].p, ptr.data(), ptr.size() * 8, hipMemcpyHostToDevice));
            if (!cols.empty()) {
                ZK_HIP(hipMemcpy(cl[w].p, cols.data(), cols.size() * 4, hipMemcpyHostToDevice));
                DevBuf raw; raw.reserve(coef.size() * 4);
                ZK_HIP(hipMemcpy(raw.p, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
                hipLaunchKernelGGL(frn_canon_to_fe_kernel, dim3(frn_blocks(cols.size())), dim3(256), 0, st, (const u32*)raw.p, (u32*)cf[w].p, (u64)cols.size());
                ZK_HIP(hipGetLastError());
                ZK_HIP(hipStreamSynchronize(st));
            }
        }
        // density trackers (prover.rs eval()): A counts auxiliaries only, B counts inputs and auxiliaries
        std::vector<int> ai, bi, li;
        for (uint32_t i = 0; i < ni; ++i) ai.push_back((int)i);
        for (uint32_t j = ni; j < n_wires; ++j) if (a_aux[j]) ai.push_back((int)j);
        for (uint32_t j = 0; j < n_wires; ++j) if (b_any[j]) bi.push_back((int)j);
        na = ai.size(); nb = bi.size();
        ZK_REQUIRE(pk.ic.n == ni, "groth16: proving key has " + std::to_string(pk.ic.n) + " public-input bases, the circuit has " + std::to_string(ni) + " inputs");
        ZK_REQUIRE(pk.l.n == n_aux, "groth16: proving key `l` query has " + std::to_string(pk.l.n) + " bases, the circuit has " + std::to_string(n_aux) + " auxiliary wires");
        ZK_REQUIRE(pk.a.n == na, "groth16: proving key `a` query has " + std::to_string(pk.a.n) + " bases, the circuit's A density is " + std::to_string(na));
        ZK_REQUIRE(pk.b_g1.n == nb && pk.b_g2.n == nb, "groth16: proving key `b` queries do not match the circuit's B density " + std::to_string(nb));
        ZK_REQUIRE(pk.h.n + 1 >= m, "groth16: proving key `h` query has " + std::to_string(pk.h.n) + " bases, the domain needs " + std::to_string(m - 1));
        nh = m - 1;
        for (uint32_t j = 0; j < n_aux; ++j) li.push_back(pk.l.inf[j] ? -1 : (int)(ni + j));
        auto upload_idx = [&](DevBuf& d, const std::vector<int>& v) {
            d.reserve(v.size() * 4 + 4);
            if (!v.empty()) ZK_HIP(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
        };
        upload_idx(a_idx, ai); upload_idx(b_idx, bi); upload_idx(l_idx, li);
        // bases: canonical coordinates -> Montgomery on the device; a point at infinity (only `l` may hold one: a wire
        // no row mentions) gets a valid stand-in and a zero scalar through l_idx
        ZK_REQUIRE(!pk.vk[0].inf[0], "groth16: alpha_g1 is the point at infinity");
        auto upload_pts = [&](DevBuf& d, const g16::PointVec& v, size_t pw, u64 count, bool allow_inf, const char* what) {
            d.reserve(std::max<size_t>(count, 1) * pw * 4);
            if (count == 0) return;
            std::vector<u32> tmp(v.w.begin(), v.w.begin() + count * pw);
            for (u64 i = 0; i < count; ++i)
                if (v.inf[i]) {
                    ZK_REQUIRE(allow_inf, std::string("groth16: proving key `") + what + "` query holds the point at infinity");
                    std::copy(pk.vk[0].w.begin(), pk.vk[0].w.begin() + pw, tmp.begin() + i * pw);
                }
            ZK_HIP(hipMemcpy(d.p, tmp.data(), count * pw * 4, hipMemcpyHostToDevice));
            G16_FQ_TO_MONT(d.p, count * pw / G16_CW, st);
        };
        upload_pts(h, pk.h, P1, nh, false, "h");
        upload_pts(l, pk.l, P1, n_aux, true, "l");
        upload_pts(a, pk.a, P1, na, false, "a");
        upload_pts(b1, pk.b_g1, P1, nb, false, "b_g1");
        upload_pts(b2, pk.b_g2, P2, nb, false, "b_g2");
        auto to_mont_host = [&](const g16::PointVec& v, size_t pw, u64 i, const char* what) {
            ZK_REQUIRE(!v.inf[i], std::string("groth16: verifying key element ") + what + " is the point at infinity");
            DevBuf d; d.reserve(pw * 4);
            ZK_HIP(hipMemcpy(d.p, v.w.data() + i * pw, pw * 4, hipMemcpyHostToDevice));
            G16_FQ_TO_MONT(d.p, pw / G16_CW, st);
            std::vector<u32> o(pw);
            ZK_HIP(hipMemcpy(o.data(), d.p, pw * 4, hipMemcpyDeviceToHost));
            return o;
        };
        alpha1 = to_mont_host(pk.vk[0], P1, 0, "alpha_g1"); beta1 = to_mont_host(pk.vk[1], P1, 0, "beta_g1");
        beta2 = to_mont_host(pk.vk[2], P2, 0, "beta_g2"); delta1 = to_mont_host(pk.vk[4], P1, 0, "delta_g1");
        delta2 = to_mont_host(pk.vk[5], P2, 0, "delta_g2");
        ZK_HIP(hipStreamSynchronize(st));
        (void)frn_domain(logm, st);
    }

    struct Term { const u32* pt; bool inf; u32 k[8]; };
    // sum of a handful of terms through the same multi-scalar kernels
    std::vector<u32> small_sum(bool g2, const std::vector<Term>& terms, bool* inf_out, hipStream_t st) const {
        const size_t pw = g2 ? P2 : P1;
        std::vector<u32> bases, sc;
        for (const Term& t : terms) {
            bool zero = true;
            for (int i = 0; i < 8; ++i) zero = zero && t.k[i] == 0;
            if (t.inf || zero) continue;
            bases.insert(bases.end(), t.pt, t.pt + pw);
            sc.insert(sc.end(), t.k, t.k + 8);
        }
        std::vector<u32> out(pw + 1, 0);
        if (sc.empty()) { *inf_out = true; return out; }
        DevBuf db, ds, dout;
        db.reserve(bases.size() * 4); ds.reserve(sc.size() * 4); dout.reserve((pw + 1) * 4);
        ZK_HIP(hipMemcpy(db.p, bases.data(), bases.size() * 4, hipMemcpyHostToDevice));
        ZK_HIP(hipMemcpy(ds.p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        if (g2) G16_MSM_G2(db.p, ds.p, sc.size() / 8, dout.p, st); else G16_MSM_G1(db.p, ds.p, sc.size() / 8, dout.p, st);
        ZK_HIP(hipStreamSynchronize(st));
        ZK_HIP(hipMemcpy(out.data(), dout.p, (pw + 1) * 4, hipMemcpyDeviceToHost));
        *inf_out = out[pw] != 0;
        return out;
    }

    uint32_t num_wires() const override { return n_wires; }
    uint32_t num_inputs() const override { return ni; }
    uint32_t domain_log() const override { return (uint32_t)logm; }

    void prove(const void* witness, bool on_device, const u64 r_[4], const u64 s_[4], u32* proof_out, std::string* json, u64* d_h_out) override {
        hipStream_t st = nullptr;
        const FrDomain& D = frn_domain(logm, st);
        DevBuf wit_c, wit_fe;
        const u32* d_wit = (const u32*)witness;
        if (!on_device) {
            wit_c.reserve((size_t)n_wires * 32);
            ZK_HIP(hipMemcpyAsync(wit_c.p, witness, (size_t)n_wires * 32, hipMemcpyHostToDevice, st));
            d_wit = (const u32*)wit_c.p;
        }
        wit_fe.reserve((size_t)n_wires * NR * 4);
        hipLaunchKernelGGL(frn_canon_to_fe_kernel, dim3(frn_blocks(n_wires)), dim3(256), 0, st, d_wit, (u32*)wit_fe.p, (u64)n_wires);
        // a_i, b_i, c_i per row (ProvingAssignment::enforce), zero-padded to the domain
        DevBuf ev[6];
        for (auto& e : ev) e.reserve(m * NR * 4);
        for (int w = 0; w < 3; ++w)
            hipLaunchKernelGGL(frn_r1cs_eval_kernel, dim3(frn_blocks(m)), dim3(256), 0, st, (const u64*)rp[w].p, (const u32*)cl[w].p, (const u32*)cf[w].p,
                               (const u32*)wit_fe.p, n_rows, (u32*)ev[w].p, m);
        ZK_HIP(hipGetLastError());
        u32* hq = frn_quotient(D, (u32*)ev[0].p, (u32*)ev[1].p, (u32*)ev[2].p, (u32*)ev[3].p, (u32*)ev[4].p, (u32*)ev[5].p, st);
        DevBuf hs, sl, sa, sb, o_h, o_l, o_a, o_b1, o_b2;
        hs.reserve(std::max<u64>(nh, 1) * 32);
        if (nh) hipLaunchKernelGGL(frn_to_canon_kernel, dim3(frn_blocks(nh)), dim3(256), 0, st, (const u32*)hq, (u32*)hs.p, m, nh);
        if (d_h_out && nh) ZK_HIP(hipMemcpyAsync(d_h_out, hs.p, nh * 32, hipMemcpyDeviceToDevice, st));
        auto gather = [&](DevBuf& d, const DevBuf& idx, u64 n) {
            d.reserve(std::max<u64>(n, 1) * 32);
            if (n) hipLaunchKernelGGL(frn_gather_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, d_wit, (const int*)idx.p, n, (u32*)d.p);
        };
        gather(sl, l_idx, n_aux); gather(sa, a_idx, na); gather(sb, b_idx, nb);
        ZK_HIP(hipGetLastError());
        auto run = [&](bool g2, const DevBuf& bases, const DevBuf& sc, u64 n, DevBuf& out) {
            const size_t pw = g2 ? P2 : P1;
            out.reserve((pw + 1) * 4);
            if (n == 0) { std::vector<u32> z(pw + 1, 0); z[pw] = 1; ZK_HIP(hipMemcpyAsync(out.p, z.data(), (pw + 1) * 4, hipMemcpyHostToDevice, st)); ZK_HIP(hipStreamSynchronize(st)); return; }
            if (g2) G16_MSM_G2(bases.p, sc.p, n, out.p, st); else G16_MSM_G1(bases.p, sc.p, n, out.p, st);
        };
        run(false, h, hs, nh, o_h); run(false, l, sl, n_aux, o_l); run(false, a, sa, na, o_a); run(false, b1, sb, nb, o_b1); run(true, b2, sb, nb, o_b2);
        ZK_HIP(hipStreamSynchronize(st));
        std::vector<u32> r_h(P1 + 1), r_l(P1 + 1), r_a(P1 + 1), r_b1(P1 + 1), r_b2(P2 + 1);
        ZK_HIP(hipMemcpy(r_h.data(), o_h.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(r_l.data(), o_l.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(r_a.data(), o_a.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(r_b1.data(), o_b1.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(r_b2.data(), o_b2.p, (P2 + 1) * 4, hipMemcpyDeviceToHost));
        // prover.rs: g_a = delta r + alpha + a;  g_b = delta2 s + beta2 + b2;
        // g_c = delta rs + alpha s + beta1 r + a s + b1 r + h + l  =  h + l + g_a s + (beta1 + b1) r
        auto term = [](const u32* pt, bool inf, const u64 k[4]) { Term t; t.pt = pt; t.inf = inf; std::memcpy(t.k, k, 32); return t; };
        const u64 one[4] = {1, 0, 0, 0};
        bool ia = false, ib = false, ic_ = false;
        std::vector<u32> A = small_sum(false, {term(delta1.data(), false, r_), term(alpha1.data(), false, one), term(r_a.data(), r_a[P1] != 0, one)}, &ia, st);
        std::vector<u32> B = small_sum(true, {term(delta2.data(), false, s_), term(beta2.data(), false, one), term(r_b2.data(), r_b2[P2] != 0, one)}, &ib, st);
        std::vector<u32> Cc = small_sum(false, {term(r_h.data(), r_h[P1] != 0, one), term(r_l.data(), r_l[P1] != 0, one), term(A.data(), ia, s_),
                                               term(beta1.data(), false, r_), term(r_b1.data(), r_b1[P1] != 0, r_)}, &ic_, st);
        ZK_REQUIRE(!ia && !ib && !ic_, "groth16: a proof element is the point at infinity");
        if (proof_out) {
            std::memcpy(proof_out, A.data(), P1 * 4);
            std::memcpy(proof_out + P1, B.data(), P2 * 4);
            std::memcpy(proof_out + P1 + P2, Cc.data(), P1 * 4);
        }
        if (json) {
            std::vector<u32> all(2 * P1 + P2);
            std::memcpy(all.data(), A.data(), P1 * 4); std::memcpy(all.data() + P1, B.data(), P2 * 4); std::memcpy(all.data() + P1 + P2, Cc.data(), P1 * 4);
            DevBuf d; d.reserve(all.size() * 4);
            ZK_HIP(hipMemcpy(d.p, all.data(), all.size() * 4, hipMemcpyHostToDevice));
            G16_FQ_TO_CANON(d.p, all.size() / G16_CW, st);
            ZK_HIP(hipStreamSynchronize(st));
            ZK_HIP(hipMemcpy(all.data(), d.p, all.size() * 4, hipMemcpyDeviceToHost));
            auto dec = [&](size_t i) { return "\"" + g16::words_to_dec(all.data() + i * G16_CW, G16_CW) + "\""; };
            // json_utils.rs:305-315 serialize_proof (to_hex = false); G2 coordinates as [c0, c1] (json_utils.rs:153-161)
            *json = "{\"pi_a\":{\"x\":" + dec(0) + ",\"y\":" + dec(1) + "},\"pi_b\":{\"x\":[" + dec(2) + "," + dec(3) + "],\"y\":[" + dec(4) + "," + dec(5) +
                    "]},\"pi_c\":{\"x\":" + dec(6) + ",\"y\":" + dec(7) + "},\"protocol\":\"groth16\",\"curve\":\"" G16_JSON_CURVE "\"}";
        }
    }
};

Groth16Setup* G16_FN(setup_new)(const g16::R1cs& rc, const g16::Params& pk) { return new G16_FN(SetupImpl)(rc, pk); }

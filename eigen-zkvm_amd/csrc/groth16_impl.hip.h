// The Groth16 prover around the multi-scalar sums, generic over the curve; included inside the scalar field's
// namespace after frntt_impl.hip.h with
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
    // One base array per group, laid out so that the proof is three multi-scalar sums with no scalar
    // multiplications left over (prover.rs's g_a, g_b, g_c with the blinding terms folded in as extra bases):
    //   G1: [ h (nh) | l (n_aux) | b_g1 (nb) | beta_g1 | a (na) | delta_g1 | alpha_g1 ]
    //        g_a = sum over the tail [a | delta | alpha] of (w_a, r, 1)
    //        g_c = sum over everything of (h, w_l, r w_b, r, s w_a, s r, s)
    //   G2: [ b_g2 (nb) | delta_g2 | beta_g2 ],  g_b = sum of (w_b, s, 1)
    DevBuf g1b, g2b;
    // the key never changes: both arrays are expanded once into window tables (2^(16 w) P for the 16 windows, msm_impl.hip.h), so a
    // proof's sums need no doublings; keys too large for the 24-bit point index of the sort keep the plain arrays
    DevBuf g1t, g2t;
    bool tables = false;
    u64 off_l = 0, off_b = 0, off_a = 0, n_g1 = 0;

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
            h2d_sync(rp[w].p, ptr.data(), ptr.size() * 8);
            if (!cols.empty()) {
                h2d_sync(cl[w].p, cols.data(), cols.size() * 4);
                DevBuf raw; raw.reserve(coef.size() * 4);
                h2d_sync(raw.p, coef.data(), coef.size() * 4);
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
            if (!v.empty()) h2d_sync(d.p, v.data(), v.size() * 4);
        };
        upload_idx(a_idx, ai); upload_idx(b_idx, bi); upload_idx(l_idx, li);
        // bases: canonical coordinates -> Montgomery on the device; a point at infinity (only `l` may hold one: a wire
        // no row mentions) gets a valid stand-in and a zero scalar through l_idx
        ZK_REQUIRE(!pk.vk[0].inf[0], "groth16: alpha_g1 is the point at infinity");
        off_l = nh; off_b = off_l + n_aux; off_a = off_b + nb + 1; n_g1 = off_a + na + 2;
        g1b.reserve(n_g1 * P1 * 4); g2b.reserve((nb + 2) * P2 * 4);
        auto put = [&](DevBuf& d, u64 at, const g16::PointVec& v, size_t pw, u64 count, bool allow_inf, const char* what) {
            if (count == 0) return;
            std::vector<u32> tmp(v.w.begin(), v.w.begin() + count * pw);
            for (u64 i = 0; i < count; ++i)
                if (v.inf[i]) {
                    ZK_REQUIRE(allow_inf, std::string("groth16: key element `") + what + "` is the point at infinity");
                    std::copy(pk.vk[0].w.begin(), pk.vk[0].w.begin() + P1, tmp.begin() + i * pw);   // stand-in (G1 only), zero scalar via l_idx
                }
            h2d_sync((u32*)d.p + at * pw, tmp.data(), count * pw * 4);
        };
        put(g1b, 0, pk.h, P1, nh, false, "h");
        put(g1b, off_l, pk.l, P1, n_aux, true, "l");
        put(g1b, off_b, pk.b_g1, P1, nb, false, "b_g1");
        put(g1b, off_b + nb, pk.vk[1], P1, 1, false, "beta_g1");
        put(g1b, off_a, pk.a, P1, na, false, "a");
        put(g1b, off_a + na, pk.vk[4], P1, 1, false, "delta_g1");
        put(g1b, off_a + na + 1, pk.vk[0], P1, 1, false, "alpha_g1");
        put(g2b, 0, pk.b_g2, P2, nb, false, "b_g2");
        put(g2b, nb, pk.vk[5], P2, 1, false, "delta_g2");
        put(g2b, nb + 1, pk.vk[2], P2, 1, false, "beta_g2");
        G16_FQ_TO_MONT(g1b.p, n_g1 * P1 / G16_CW, st);
        G16_FQ_TO_MONT(g2b.p, (nb + 2) * P2 / G16_CW, st);
        tables = n_g1 < (1ull << 24) && !getenv("ZK_GROTH16_NO_TABLES");   // the knob keeps the plain-array path testable at small sizes
        if (tables) {
            g1t.reserve(G16_MSM_G1_TABLE_BYTES(n_g1)); g2t.reserve(G16_MSM_G2_TABLE_BYTES(nb + 2));
            G16_MSM_G1_PREPARE(g1b.p, n_g1, g1t.p, st);
            G16_MSM_G2_PREPARE(g2b.p, nb + 2, g2t.p, st);
            ZK_HIP(hipStreamSynchronize(st));
            g1b.release(); g2b.release();
        }
        ZK_HIP(hipStreamSynchronize(st));
        (void)frn_domain(logm, st);
    }

    uint32_t num_wires() const override { return n_wires; }
    uint32_t num_inputs() const override { return ni; }
    uint32_t domain_log() const override { return (uint32_t)logm; }

    // Three independent chains share the device: g_b (G2, the longest) and g_a start as soon as the witness is
    // resident; row evaluations, the quotient and g_c run beside them.  Each multi-scalar sum ends in a short
    // latency-bound tail (bucket hierarchy + window Horner on a few waves) that the other chains' bulk work hides.
    hipStream_t streams[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_ready = nullptr;
    int device = -1;
    ~G16_FN(SetupImpl)() override {
        for (auto& st : streams) if (st) { forget_stream(st); (void)hipStreamDestroy(st); }
        if (ev_ready) (void)hipEventDestroy(ev_ready);
    }

    void prove(const void* witness, bool on_device, const u64 r_[4], const u64 s_[4], u32* proof_out, std::string* json, u64* d_h_out) override {
        if (!streams[0]) {
            ZK_HIP(hipGetDevice(&device));
            for (auto& st : streams) ZK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            ZK_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        }
        on_stream(streams[1]); on_stream(streams[2]);                    // all three are registered with the pool;
        hipStream_t st = on_stream(streams[0]);                          // this thread issues on streams[0]
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
        // scalars of the three sums (see the base layout above)
        DevBuf scC, scA, scB, blind, o_a, o_b, o_c;
        scC.reserve(n_g1 * 32); scA.reserve((na + 2) * 32); scB.reserve((nb + 2) * 32); blind.reserve(64 + 2 * NR * 4);
        o_a.reserve((P1 + 1) * 4); o_b.reserve((P2 + 1) * 4); o_c.reserve((P1 + 1) * 4);
        u32* C0 = (u32*)scC.p;
        ZK_HIP(hipMemcpyAsync(blind.p, r_, 32, hipMemcpyHostToDevice, st));
        ZK_HIP(hipMemcpyAsync((u32*)blind.p + 8, s_, 32, hipMemcpyHostToDevice, st));
        fe* rs_fe = (fe*)((u32*)blind.p + 16);
        hipLaunchKernelGGL(frn_blind_kernel, dim3(1), dim3(64), 0, st, (const u32*)blind.p, rs_fe, (u32*)scA.p + na * 8, (u32*)scB.p + nb * 8,
                           C0 + (off_b + nb) * 8, C0 + (off_a + na) * 8);
        auto gather = [&](u32* out, const DevBuf& idx, u64 n, const fe* scale) {
            if (n) hipLaunchKernelGGL(frn_gather_kernel, dim3(frn_blocks(n)), dim3(256), 0, st, d_wit, (const u32*)wit_fe.p, (const int*)idx.p, n, scale, out);
        };
        gather((u32*)scB.p, b_idx, nb, nullptr);
        gather((u32*)scA.p, a_idx, na, nullptr);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipEventRecord(ev_ready, st));
        std::exception_ptr err[2];
        auto side = [&](int k) {
            try {
                ZK_HIP(hipSetDevice(device));
                on_stream(streams[k + 1]);                                  // this thread's allocations are ordered on its own stream
                ZK_HIP(hipStreamWaitEvent(streams[k + 1], ev_ready, 0));
                if (k == 0) { if (tables) G16_MSM_G2_FIXED(g2t.p, nb + 2, 0, scB.p, nb + 2, o_b.p, streams[1]); else G16_MSM_G2(g2b.p, scB.p, nb + 2, o_b.p, streams[1]); }
                else { if (tables) G16_MSM_G1_FIXED(g1t.p, n_g1, off_a, scA.p, na + 2, o_a.p, streams[2]); else G16_MSM_G1((const u32*)g1b.p + off_a * P1, scA.p, na + 2, o_a.p, streams[2]); }
            } catch (...) { err[k] = std::current_exception(); }
        };
        std::thread tb(side, 0), ta(side, 1);
        std::exception_ptr main_err;
        try {
            // a_i, b_i, c_i per row (ProvingAssignment::enforce), zero-padded to the domain
            DevBuf ev[6];
            for (auto& e : ev) e.reserve(m * NR * 4);
            for (int w = 0; w < 3; ++w)
                hipLaunchKernelGGL(frn_r1cs_eval_kernel, dim3(frn_blocks(m)), dim3(256), 0, st, (const u64*)rp[w].p, (const u32*)cl[w].p, (const u32*)cf[w].p,
                                   (const u32*)wit_fe.p, n_rows, (u32*)ev[w].p, m);
            ZK_HIP(hipGetLastError());
            u32* hq = frn_quotient(D, (u32*)ev[0].p, (u32*)ev[1].p, (u32*)ev[2].p, (u32*)ev[3].p, (u32*)ev[4].p, (u32*)ev[5].p, st);
            if (nh) hipLaunchKernelGGL(frn_to_canon_kernel, dim3(frn_blocks(nh)), dim3(256), 0, st, (const u32*)hq, C0, m, nh);
            if (d_h_out && nh) ZK_HIP(hipMemcpyAsync(d_h_out, C0, nh * 32, hipMemcpyDeviceToDevice, st));
            gather(C0 + off_l * 8, l_idx, n_aux, nullptr);
            gather(C0 + off_b * 8, b_idx, nb, rs_fe);            // r w_b
            gather(C0 + off_a * 8, a_idx, na, rs_fe + 1);        // s w_a
            ZK_HIP(hipGetLastError());
            if (tables) G16_MSM_G1_FIXED(g1t.p, n_g1, 0, scC.p, n_g1, o_c.p, st); else G16_MSM_G1(g1b.p, scC.p, n_g1, o_c.p, st);
            ZK_HIP(hipStreamSynchronize(st));                     // ev[] goes back to the pool
        } catch (...) { main_err = std::current_exception(); }
        tb.join(); ta.join();
        for (auto& s2 : streams) (void)hipStreamSynchronize(s2);
        if (main_err) std::rethrow_exception(main_err);
        for (auto& e : err) if (e) std::rethrow_exception(e);
        std::vector<u32> A(P1 + 1), B(P2 + 1), Cc(P1 + 1);
        ZK_HIP(hipMemcpy(A.data(), o_a.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(B.data(), o_b.p, (P2 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_HIP(hipMemcpy(Cc.data(), o_c.p, (P1 + 1) * 4, hipMemcpyDeviceToHost));
        ZK_REQUIRE(!A[P1] && !B[P2] && !Cc[P1], "groth16: a proof element is the point at infinity");
        if (proof_out) {
            std::memcpy(proof_out, A.data(), P1 * 4);
            std::memcpy(proof_out + P1, B.data(), P2 * 4);
            std::memcpy(proof_out + P1 + P2, Cc.data(), P1 * 4);
        }
        if (json) {
            std::vector<u32> all(2 * P1 + P2);
            std::memcpy(all.data(), A.data(), P1 * 4); std::memcpy(all.data() + P1, B.data(), P2 * 4); std::memcpy(all.data() + P1 + P2, Cc.data(), P1 * 4);
            DevBuf d; d.reserve(all.size() * 4);
            h2d_sync(d.p, all.data(), all.size() * 4);
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

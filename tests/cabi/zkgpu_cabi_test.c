/* A plain-C client of include/zkgpu.h, built with `gcc -std=c99 -I include` and linked against libzkgpu.so: the header must be
 * valid C, every prototype used here must agree with what the library exports, and the calls must behave as the header says.
 * (tests/test_cabi_c_client.py builds it; `link` mode runs without a GPU, `run` mode on the GPU box.)
 *
 *   zkgpu_cabi_test link
 *   zkgpu_cabi_test run <starkinfo_program.json> <starkStruct.json> <pols.const> <pols.cm> <out zkin.json>
 */
#include "zkgpu.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)n + 1);
    if (fread(b, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    b[n] = 0; fclose(f);
    if (len) *len = (size_t)n;
    return b;
}
#define CHECK(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, zk_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "link") == 0) {
        /* take the address of one entry point per family: an undefined or mistyped symbol fails the build, not the run */
        void *fns[] = {(void *)zk_init, (void *)zk_gl_ntt, (void *)zk_gl_lde, (void *)zk_gl_ntt_dev, (void *)zk_gl_lde_dev,
                       (void *)zk_gl_poseidon, (void *)zk_gl_linearhash, (void *)zk_gl_merkelize, (void *)zk_merkle_root,
                       (void *)zk_merkle_group_proof, (void *)zk_merkle_free, (void *)zk_transcript_new, (void *)zk_fri_fold_dev,
                       (void *)zk_stark_calculate_z_dev, (void *)zk_msm_g1_bn254, (void *)zk_stark_setup_new, (void *)zk_stark_gen,
                       (void *)zk_stark_gen_dev, (void *)zk_stark_gen_dev_on, (void *)zk_program_run_rows_dev, (void *)zk_string_free, (void *)zk_stark_setup_free, (void *)zk_c12_exec_new,
                       (void *)zk_groth16_setup_new, (void *)zk_program_compile, (void *)zk_stark_new, (void *)zk_stark_eval, (void *)zk_stark_commit_stage,
                       (void *)zk_stark_set_challenge, (void *)zk_stark_challenge, (void *)zk_stark_evals, (void *)zk_stark_fri_prove, (void *)zk_stark_finish, (void *)zk_stark_free,
                       (void *)zk_fri_prove_dev, (void *)zk_stark_verify_set_reference_compat};
        unsigned n = 0;
        for (unsigned i = 0; i < sizeof fns / sizeof fns[0]; ++i) n += fns[i] != NULL;
        printf("linked %u entry points; p = %llu; devices = %d\n", n, (unsigned long long)zk_gl_modulus(), zk_device_count());
        return zk_gl_modulus() == 0xFFFFFFFF00000001ULL ? 0 : 1;
    }
    if (argc != 7 || strcmp(argv[1], "run") != 0) { fprintf(stderr, "usage: see the head of this file\n"); return 2; }
    if (zk_device_count() < 1) { fprintf(stderr, "no GPU: the library has no CPU fallback\n"); return 3; }
    CHECK(zk_init(0));

    /* fft_p::fft / ifft through the host-pointer entry points: inverse(forward(x)) == x, 3 columns of 2^12 */
    enum { NB = 12, W = 3 };
    const size_t n = ((size_t)1 << NB) * W;
    uint64_t *x = (uint64_t *)malloc(n * 8), *y = (uint64_t *)malloc(n * 8), *z = (uint64_t *)malloc(n * 8);
    for (size_t i = 0; i < n; ++i) x[i] = (i * 0x9E3779B97F4A7C15ULL + 12345) % 0xFFFFFFFF00000001ULL;
    CHECK(zk_gl_ntt(x, y, W, NB, 0));
    CHECK(zk_gl_ntt(y, z, W, NB, 1));
    if (memcmp(x, z, n * 8) != 0) { fprintf(stderr, "NTT round trip differs\n"); return 1; }
    if (zk_gl_ntt(x, x, W, NB, 0) == 0) { fprintf(stderr, "aliased src/dst must be rejected\n"); return 1; }
    printf("ntt round trip ok; aliasing rejected: %s\n", zk_last_error());

    /* StarkSetup's constant tree by hand: fft_p::interpolate + MerkleTreeGL::merkelize + root */
    size_t const_bytes, cm_bytes;
    uint64_t *cst = (uint64_t *)slurp(argv[4], &const_bytes), *cm = (uint64_t *)slurp(argv[5], &cm_bytes);
    char *prog = slurp(argv[2], NULL), *ss = slurp(argv[3], NULL);
    const char *nb_s = strstr(ss, "\"nBits\""), *ne_s = strstr(ss, "\"nBitsExt\"");
    const unsigned nbits = (unsigned)atoi(strchr(nb_s, ':') + 1), nbits_ext = (unsigned)atoi(strchr(ne_s, ':') + 1);
    const unsigned n_const = (unsigned)(const_bytes / 8 >> nbits);
    uint64_t *ext = (uint64_t *)malloc(((size_t)n_const << nbits_ext) * 8);
    CHECK(zk_gl_lde(cst, n_const, nbits, ext, nbits_ext));
    zk_merkle_t *tree = zk_gl_merkelize(ext, n_const, (uint64_t)1 << nbits_ext);
    if (!tree) { fprintf(stderr, "merkelize: %s\n", zk_last_error()); return 1; }
    uint64_t root[4], root2[4];
    CHECK(zk_merkle_root(tree, root));
    if (zk_merkle_group_proof(tree, (uint64_t)1 << nbits_ext, ext, ext) == 0) { fprintf(stderr, "idx >= height must be an error\n"); return 1; }
    CHECK(zk_merkle_free(tree));
    printf("const_root %llu %llu %llu %llu\n", (unsigned long long)root[0], (unsigned long long)root[1], (unsigned long long)root[2], (unsigned long long)root[3]);

    /* the whole prover: StarkSetup::new + StarkProof::stark_gen */
    zk_stark_setup_t *su = zk_stark_setup_new(prog, ss, cst, const_bytes / 8);
    if (!su) { fprintf(stderr, "zk_stark_setup_new: %s\n", zk_last_error()); return 1; }
    CHECK(zk_stark_setup_const_root(su, root2));
    if (memcmp(root, root2, 32) != 0) { fprintf(stderr, "setup's constant root differs from the hand-built tree\n"); return 1; }
    char *zkin = zk_stark_gen(su, cm, cm_bytes / 8);
    if (!zkin) { fprintf(stderr, "zk_stark_gen: %s\n", zk_last_error()); return 1; }
    FILE *o = fopen(argv[6], "w"); fputs(zkin, o); fclose(o);
    printf("proof written: %zu bytes\n", strlen(zkin));
    /* stark_verify on the prover's own output (prove.rs:124-132): accepted; with another constant root: rejected, not an error */
    if (zk_stark_verify(su, zkin) != 1) { fprintf(stderr, "zk_stark_verify: %s\n", zk_last_error()); return 1; }
    root2[0] ^= 1;
    if (zk_stark_verify_with(prog, ss, root2, zkin) != 0) { fprintf(stderr, "a proof against another constant root must be rejected\n"); return 1; }
    root2[0] ^= 1;
    if (zk_stark_verify_with(prog, ss, root2, zkin) != 1) { fprintf(stderr, "zk_stark_verify_with: %s\n", zk_last_error()); return 1; }
    if (zk_stark_verify(su, "{") != -1) { fprintf(stderr, "malformed zkin must be an error\n"); return 1; }
    CHECK(zk_stark_setup_set_self_check(su, 1));
    char *zkin2 = zk_stark_gen(su, cm, cm_bytes / 8);
    if (!zkin2 || strcmp(zkin, zkin2) != 0) { fprintf(stderr, "self-checked proof differs: %s\n", zk_last_error()); return 1; }
    zk_string_free(zkin2);
    printf("verified: accepted, rejected under another constant root, self check on\n");
    /* the same proof through the staged seams (zkgpu.h "the staged prover"; stark_gen.rs:279-545 in the reference's order): byte-equal */
    {
        zk_stark_ctx_t *c = zk_stark_new(su, cm, NULL, cm_bytes / 8, NULL);
        if (!c) { fprintf(stderr, "zk_stark_new: %s\n", zk_last_error()); return 1; }
        uint64_t r[4], ch[3];
        if (zk_stark_eval(c, ZK_STEP_2PREV) == 0) { fprintf(stderr, "step2prev before the first commitment must be refused\n"); return 1; }
        CHECK(zk_stark_commit_stage(c, 1, r)); CHECK(zk_stark_challenge(c, 0, ch)); CHECK(zk_stark_challenge(c, 1, NULL));
        CHECK(zk_stark_eval(c, ZK_STEP_2PREV)); CHECK(zk_stark_calculate_h1h2(c));
        CHECK(zk_stark_commit_stage(c, 2, NULL)); CHECK(zk_stark_challenge(c, 2, NULL)); CHECK(zk_stark_challenge(c, 3, NULL));
        CHECK(zk_stark_eval(c, ZK_STEP_3PREV)); CHECK(zk_stark_calculate_z(c)); CHECK(zk_stark_eval(c, ZK_STEP_3));
        CHECK(zk_stark_commit_stage(c, 3, NULL)); CHECK(zk_stark_challenge(c, 4, NULL));
        CHECK(zk_stark_eval(c, ZK_STEP_42NS));
        CHECK(zk_stark_commit_stage(c, 4, NULL)); CHECK(zk_stark_challenge(c, 7, NULL));
        const int n_ev = zk_stark_evals(c, NULL, 0);
        if (n_ev < 1) { fprintf(stderr, "zk_stark_evals: %s\n", zk_last_error()); return 1; }
        CHECK(zk_stark_challenge(c, 5, NULL)); CHECK(zk_stark_challenge(c, 6, NULL));
        CHECK(zk_stark_eval(c, ZK_STEP_52NS));
        if (zk_stark_fri_pol_dev(c) == NULL || zk_stark_tree(c, 1) == NULL || zk_stark_tree(c, 5) == NULL) { fprintf(stderr, "staged accessors\n"); return 1; }
        CHECK(zk_stark_fri_prove(c));
        char *zs = zk_stark_finish(c);
        if (!zs || strcmp(zs, zkin) != 0) { fprintf(stderr, "the staged proof differs from zk_stark_gen's: %s\n", zk_last_error()); return 1; }
        if (strstr(zkin, "\"root1\"") == NULL || r[0] == 0) { fprintf(stderr, "commit_stage did not hand out the root\n"); return 1; }
        zk_string_free(zs);
        CHECK(zk_stark_free(c));
        printf("staged prover: %d evaluations, zkin byte-equal to zk_stark_gen's\n", n_ev);
    }
    zk_string_free(zkin);
    if (zk_stark_gen(su, cm, cm_bytes / 8 - 1) != NULL) { fprintf(stderr, "a short trace must be rejected\n"); return 1; }
    CHECK(zk_stark_setup_free(su));
    free(x); free(y); free(z); free(cst); free(cm); free(prog); free(ss); free(ext);
    return 0;
}

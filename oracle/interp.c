/* ORACLE -- TEST INFRASTRUCTURE ONLY (see gl.h).  C restatement of the reference's constraint interpreter so that
 * the restated prover (oracle/stark_prover.py) can run at 2^16..2^20 rows: starky/src/interpreter.rs:91-175
 * (Block::eval), :228-234 (get_i), :236-283 (get_value) with the runtime-dim arithmetic of F3G
 * (starky/src/f3g.rs:323-449).  Same program encoding as oracle/interp.py, flattened to int64 words; the two
 * are compared op for op in tests/test_oracle_interp.py.
 *
 * An operand is 6 words {kind, a, b, c, d, e}:
 *   0 tmp{id}  1 mem{buf, column, stride, dim, prime}  2 number{value}  3 public{id}  4 challenge{id}  5 eval{id}
 *   6 x  7 Zi  8 xDivXSubXi  9 xDivXSubWXi
 * An instruction is {op (0 add, 1 sub, 2 mul, 3 copy), dest[6], src0[6], src1[6]} = 19 words.
 * Rows run in order, as the reference does inside a chunk (stark_gen.rs:752-783); when no section is both read
 * and written by the program the order is unobservable and the rows are spread over the host threads.        */
#include "gl.h"
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t v[3]; int dim; } val_t;

static inline val_t v_add(val_t a, val_t b) {                       /* f3g.rs:323-361 */
    val_t r;
    if (a.dim == 3 && b.dim == 3) { r.dim = 3; for (int i = 0; i < 3; ++i) r.v[i] = gl_add(a.v[i], b.v[i]); }
    else if (a.dim == 3) { r = a; r.v[0] = gl_add(a.v[0], b.v[0]); }
    else if (b.dim == 3) { r = b; r.v[0] = gl_add(b.v[0], a.v[0]); }
    else { r.dim = 1; r.v[0] = gl_add(a.v[0], b.v[0]); r.v[1] = r.v[2] = 0; }
    return r;
}
static inline val_t v_sub(val_t a, val_t b) {                       /* f3g.rs:370-398 */
    val_t r;
    if (a.dim == 3 && b.dim == 3) { r.dim = 3; for (int i = 0; i < 3; ++i) r.v[i] = gl_sub(a.v[i], b.v[i]); }
    else if (a.dim == 3) { r = a; r.v[0] = gl_sub(a.v[0], b.v[0]); }
    else if (b.dim == 3) { r.dim = 3; r.v[0] = gl_sub(a.v[0], b.v[0]); r.v[1] = gl_neg(b.v[1]); r.v[2] = gl_neg(b.v[2]); }
    else { r.dim = 1; r.v[0] = gl_sub(a.v[0], b.v[0]); r.v[1] = r.v[2] = 0; }
    return r;
}
static inline val_t v_mul(val_t a, val_t b) {                       /* f3g.rs:407-449 */
    val_t r;
    if (a.dim == 3 && b.dim == 3) {
        f3_t x, y; memcpy(x.v, a.v, 24); memcpy(y.v, b.v, 24);
        f3_t z = f3_mul(x, y); memcpy(r.v, z.v, 24); r.dim = 3;
    } else if (a.dim == 3) { r.dim = 3; for (int i = 0; i < 3; ++i) r.v[i] = gl_mul(a.v[i], b.v[0]); }
    else if (b.dim == 3) { r.dim = 3; for (int i = 0; i < 3; ++i) r.v[i] = gl_mul(b.v[i], a.v[0]); }
    else { r.dim = 1; r.v[0] = gl_mul(a.v[0], b.v[0]); r.v[1] = r.v[2] = 0; }
    return r;
}

typedef struct {
    uint64_t **bufs; uint64_t n, next;
    const uint64_t *publics, *challenges, *evals, *x, *zi; uint64_t zi_len;
    const uint64_t *xdiv, *xdivw;
} env_t;

static inline val_t one(uint64_t a) { val_t r = {{a, 0, 0}, 1}; return r; }
static inline val_t three(const uint64_t *p) { val_t r = {{p[0], p[1], p[2]}, 3}; return r; }

static inline val_t get(const int64_t *o, uint64_t i, const val_t *tmp, const env_t *e) {
    switch (o[0]) {
    case 0: return tmp[o[1]];
    case 1: {
        uint64_t row = (i + (o[5] ? e->next : 0)) % e->n;           /* interpreter.rs:228-234 */
        const uint64_t *p = e->bufs[o[1]] + (uint64_t)o[2] + row * (uint64_t)o[3];
        return o[4] == 1 ? one(p[0]) : three(p);
    }
    case 2: return one((uint64_t)o[1]);
    case 3: return one(e->publics[o[1]]);
    case 4: return three(e->challenges + 3 * o[1]);
    case 5: return three(e->evals + 3 * o[1]);
    case 6: return one(e->x[i]);
    case 7: return one(e->zi[i % e->zi_len]);
    case 8: return three(e->xdiv + 3 * i);
    default: return three(e->xdivw + 3 * i);
    }
}

static void run_row(const int64_t *code, uint64_t n_ops, uint64_t i, val_t *tmp, const env_t *e) {
    for (uint64_t k = 0; k < n_ops; ++k) {
        const int64_t *c = code + 19 * k;
        val_t a = get(c + 7, i, tmp, e), r;
        if (c[0] == 3) r = a;
        else {
            val_t b = get(c + 13, i, tmp, e);
            r = c[0] == 0 ? v_add(a, b) : c[0] == 1 ? v_sub(a, b) : v_mul(a, b);
        }
        const int64_t *d = c + 1;
        if (d[0] == 0) tmp[d[1]] = r;
        else {                                                         /* interpreter.rs:143-166: dim-3 results unpack into 3 cells */
            uint64_t row = (i + (d[5] ? e->next : 0)) % e->n;
            uint64_t *p = e->bufs[d[1]] + (uint64_t)d[2] + row * (uint64_t)d[3];
            for (int j = 0; j < r.dim; ++j) p[j] = r.v[j];
        }
    }
}

/* returns 0, or -1 for a malformed program */
int orc_interp_run(const int64_t *code, uint64_t n_ops, uint64_t n_tmp, uint64_t **bufs, uint64_t n_bufs, uint64_t n, uint64_t next,
                   const uint64_t *publics, const uint64_t *challenges, const uint64_t *evals, const uint64_t *x,
                   const uint64_t *zi, uint64_t zi_len, const uint64_t *xdiv, const uint64_t *xdivw) {
    env_t e = {bufs, n, next, publics, challenges, evals, x, zi, zi_len ? zi_len : 1, xdiv, xdivw};
    /* a section that is written and also read makes the row order observable */
    char *written = (char *)calloc(n_bufs + 1, 1), *read = (char *)calloc(n_bufs + 1, 1);
    for (uint64_t k = 0; k < n_ops; ++k) {
        const int64_t *c = code + 19 * k;
        if (c[0] < 0 || c[0] > 3) { free(written); free(read); return -1; }
        for (int s = 0; s < 3; ++s) {
            const int64_t *o = c + 1 + 6 * s;
            if (o[0] == 0 && (uint64_t)o[1] >= n_tmp) { free(written); free(read); return -1; }
            if (o[0] == 1) {
                if ((uint64_t)o[1] >= n_bufs) { free(written); free(read); return -1; }
                (s == 0 ? written : read)[o[1]] = 1;
            }
        }
    }
    int ordered = 0;
    for (uint64_t b = 0; b < n_bufs; ++b) ordered |= written[b] && read[b];
    free(written); free(read);
    if (ordered) {
        val_t *tmp = (val_t *)calloc(n_tmp + 1, sizeof(val_t));
        for (uint64_t i = 0; i < n; ++i) run_row(code, n_ops, i, tmp, &e);
        free(tmp);
    } else {
        #pragma omp parallel
        {
            val_t *tmp = (val_t *)calloc(n_tmp + 1, sizeof(val_t));
            #pragma omp for schedule(static)
            for (uint64_t i = 0; i < n; ++i) run_row(code, n_ops, i, tmp, &e);
            free(tmp);
        }
    }
    return 0;
}

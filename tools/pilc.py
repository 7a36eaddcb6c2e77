#!/usr/bin/env python3
"""pilc -- a small PIL -> compiled-PIL (`*.pil.json`) translator for the subset of the language the
reference's own circuits use (starkjs/{fibonacci,poseidon,permutation,plookup,connection}/*.pil).

Why it exists: the reference compiles its PIL sources with pilcom (npm `pilcom ^0.0.20`,
starkjs/package.json:21; called from starkjs/src/pil_verifier.js:28), which is not in this image, and
starky only ever consumes the compiled JSON (types.rs:134-155 `PIL`).  BASELINE config 3 / the 2^24
headline name `starkjs/poseidon/poseidong.pil`, for which the tree holds no compiled form.  This tool
produces it.  It is pinned by the source <-> compiled pairs the reference does hold
(tests/test_pilc.py): fibonacci_old.pil <-> starky/data/fib.pil.json, permutation <-> pe.pil.json,
plookup <-> plookup.pil.json, connection <-> connection.pil.json.

Host-side input preparation only: nothing here is on the measured path and the prover never imports it.

What is modelled of pilcom's behaviour (everything the pairs above can witness):
  * ids: committed / constant columns numbered in declaration order (arrays take `len` ids), `pol x = e`
    and every identity / lookup operand push one entry on `expressions`;
  * degrees: column 1, number / public 0, add / sub max, mul sum; an intermediate polynomial (`imP`) of
    degree 2 that is referenced from another expression is reduced to degree 1 through a quotient
    (`idQ`, numbered in first-use order, depth first from the identities) and the referencing
    expression lists it in `deps`; operands of lookups/permutations are reduced the same way;
  * constant folding of number (op) number, x*1, x*0, x+0 (never x-0: identities keep `- 0`);
  * JSON key order and indentation of JSON.stringify(pil, null, 1).
"""
import json
import os
import re
import sys

P = 0xFFFFFFFF00000001

TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>0x[0-9a-fA-F]+|\d+)
  | (?P<str>"[^"]*")
  | (?P<id>[A-Za-z_][A-Za-z_0-9]*(?:\.[A-Za-z_][A-Za-z_0-9]*)?)
  | (?P<op>\*\*|[-+*()\[\]{},;=':%])
""", re.X | re.S)

KEYWORDS = {"namespace", "pol", "constant", "commit", "public", "include", "let", "in", "is", "connect", "int"}


class PilError(Exception):
    pass


def tokenize(text, fname):
    out, pos, line = [], 0, 1
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise PilError("%s:%d: unexpected character %r" % (fname, line, text[pos]))
        kind = m.lastgroup
        if kind != "ws":
            out.append((kind, m.group(), line))
        line += m.group().count("\n")
        pos = m.end()
    out.append(("eof", "", line))
    return out


class Parser:
    """statements -> list of dicts with a `type` and the source line of their first token"""

    def __init__(self, text, fname):
        self.t, self.i, self.fname = tokenize(text, fname), 0, fname

    def peek(self, k=0): return self.t[self.i + k]
    def at(self, v): return self.t[self.i][1] == v and self.t[self.i][0] in ("op", "id")

    def take(self, v=None):
        tok = self.t[self.i]
        if v is not None and tok[1] != v:
            raise PilError("%s:%d: expected %r, found %r" % (self.fname, tok[2], v, tok[1]))
        self.i += 1
        return tok

    def statements(self):
        sts = []
        while self.peek()[0] != "eof":
            sts.append(self.statement())
        return sts

    def end(self):
        if self.at(";"):
            self.take()

    def statement(self):
        kind, v, line = self.peek()
        if v == "include":
            self.take(); s = self.take()[1].strip('"'); self.end()
            return {"type": "include", "file": s, "line": line}
        if v == "let":                                         # let N: int = 2**10;
            self.take(); name = self.take()[1]
            if self.at(":"):
                self.take(); self.take()
            self.take("="); e = self.expr(); self.end()
            return {"type": "constdef", "name": name, "e": e, "line": line}
        if v == "constant" and self.peek(1)[1] == "%":        # constant %N = 2**10;
            self.take(); self.take("%"); name = self.take()[1]; self.take("="); e = self.expr(); self.end()
            return {"type": "constdef", "name": name, "e": e, "line": line}
        if v == "namespace":
            self.take(); name = self.take()[1]; self.take("("); e = self.expr(); self.take(")"); self.end()
            return {"type": "namespace", "name": name, "e": e, "line": line}
        if v == "public":
            self.take(); name = self.take()[1]; self.take("=")
            pol = self.take()[1]; idx_arr = None
            if self.at("["):
                self.take(); idx_arr = self.expr(); self.take("]")
            self.take("("); e = self.expr(); self.take(")"); self.end()
            return {"type": "public", "name": name, "pol": pol, "arr": idx_arr, "e": e, "line": line}
        if v == "pol" and self.peek(1)[1] in ("constant", "commit"):
            self.take(); what = self.take()[1]
            names = []
            while True:
                n = self.take()[1]; ln = None
                if self.at("["):
                    self.take(); ln = self.expr(); self.take("]")
                names.append((n, ln))
                if self.at(","):
                    self.take(); continue
                break
            self.end()
            return {"type": "decl", "what": what, "names": names, "line": line}
        if v == "pol":
            self.take(); name = self.take()[1]; self.take("="); e = self.expr(); self.end()
            return {"type": "poldef", "name": name, "e": e, "line": line}
        # identity, lookup, permutation or connection
        sel_f, f = self.puexpr()
        if self.at("=") and sel_f is None and f is not None and not isinstance(f, list):
            self.take(); rhs = self.expr(); self.end()
            return {"type": "identity", "a": f, "b": rhs, "line": line}
        kw = self.take()[1]
        if kw not in ("in", "is", "connect"):
            raise PilError("%s:%d: expected '=', 'in', 'is' or 'connect', found %r" % (self.fname, line, kw))
        sel_t, t = self.puexpr()
        self.end()
        as_list = lambda x: x if isinstance(x, list) else [x]
        return {"type": {"in": "plookup", "is": "permutation", "connect": "connection"}[kw], "selF": sel_f, "f": as_list(f),
                "selT": sel_t, "t": as_list(t), "line": line}

    def puexpr(self):
        """[selector] { e, e, ... }   |   e"""
        if self.at("{"):
            return None, self.elist()
        e = self.expr()
        if self.at("{"):
            return e, self.elist()
        return None, e

    def elist(self):
        self.take("{"); out = [self.expr()]
        while self.at(","):
            self.take(); out.append(self.expr())
        self.take("}")
        return out

    # expressions: + - (left assoc) < * < ** (right assoc) < unary -
    def expr(self):
        a = self.term()
        while self.at("+") or self.at("-"):
            op = "add" if self.take()[1] == "+" else "sub"
            a = {"op": op, "values": [a, self.term()]}
        return a

    def term(self):
        a = self.power()
        while self.at("*"):
            self.take()
            a = {"op": "mul", "values": [a, self.power()]}
        return a

    def power(self):
        a = self.unary()
        if self.at("**"):
            self.take()
            return {"op": "pow", "values": [a, self.power()]}
        return a

    def unary(self):
        if self.at("-"):
            self.take()
            return {"op": "neg", "values": [self.unary()]}
        kind, v, line = self.take()
        if kind == "num":
            return {"op": "number", "value": int(v, 16) if v.startswith("0x") else int(v)}
        if v == "(":
            e = self.expr(); self.take(")")
            return e
        if v == ":":
            return {"op": "public", "name": self.take()[1]}
        if v == "%":
            return {"op": "constant", "name": self.take()[1]}
        if kind == "id" and v not in KEYWORDS:
            e = {"op": "pol", "name": v, "next": False, "line": line}
            if self.at("["):
                self.take(); e["idxExp"] = self.expr(); self.take("]")
            if self.at("'"):
                self.take(); e["next"] = True
            return e
        raise PilError("%s:%d: unexpected %r in expression" % (self.fname, line, v))


class Compiler:
    def __init__(self):
        self.references, self.ref_order = {}, []
        self.publics = {}
        self.constants = {}
        self.n_cm = self.n_const = self.n_im = self.n_q = 0
        self.expressions = []
        self.pol_ids, self.plookups, self.permutations, self.connections = [], [], [], []
        self.namespace, self.pol_deg = "Global", None
        self.sources = {}                                     # basename -> text overriding the file (tests)

    # ---- pass 1: declarations ----------------------------------------------------------------
    def load(self, path, text=None):
        fname = os.path.basename(path)
        if text is None:
            text = self.sources.get(fname)
        if text is None:
            with open(path) as f:
                text = f.read()
        for s in Parser(text, fname).statements():
            t = s["type"]
            if t == "include":
                self.load(os.path.join(os.path.dirname(path), s["file"]))
            elif t == "constdef":
                self.constants[s["name"]] = self.const_eval(s["e"])
            elif t == "namespace":
                self.namespace, self.pol_deg = s["name"], self.const_eval(s["e"])
            elif t == "decl":
                for name, ln in s["names"]:
                    n = 1 if ln is None else self.const_eval(ln)
                    full = self.namespace + "." + name
                    if full in self.references:
                        raise PilError("%s:%d: %s already defined" % (fname, s["line"], full))
                    if s["what"] == "commit":
                        ref = {"type": "cmP", "id": self.n_cm}; self.n_cm += n
                    else:
                        ref = {"type": "constP", "id": self.n_const}; self.n_const += n
                    ref.update(polDeg=self.pol_deg, isArray=ln is not None)
                    if ln is not None:
                        ref["len"] = n
                    self.references[full] = ref
            elif t == "poldef":
                full = self.namespace + "." + s["name"]
                self.references[full] = {"type": "imP", "id": len(self.expressions), "polDeg": self.pol_deg, "isArray": False}
                self.expressions.append(self.scope(s["e"]))
                self.n_im += 1
            elif t == "identity":
                self.pol_ids.append({"e": len(self.expressions), "fileName": fname, "line": s["line"], "ns": self.namespace})
                self.expressions.append(self.scope({"op": "sub", "values": [s["a"], s["b"]]}))
            elif t in ("plookup", "permutation"):
                d = {"fileName": fname, "line": s["line"]}
                for side, sel in (("f", "selF"), ("t", "selT")):
                    d[side] = []
                    for e in s[side]:
                        d[side].append(len(self.expressions)); self.expressions.append(self.scope(e))
                    d[sel] = None
                    if s[sel] is not None:
                        d[sel] = len(self.expressions); self.expressions.append(self.scope(s[sel]))
                if len(d["f"]) != len(d["t"]):
                    raise PilError("%s:%d: lookup sides differ in length" % (fname, s["line"]))
                (self.plookups if t == "plookup" else self.permutations).append(d)
            elif t == "connection":
                d = {"fileName": fname, "line": s["line"], "pols": [], "connections": []}
                for side, key in (("f", "pols"), ("t", "connections")):
                    for e in s[side]:
                        d[key].append(len(self.expressions)); self.expressions.append(self.scope(e))
                self.connections.append(d)
            elif t == "public":
                ref, off = self.lookup(s["pol"], s.get("arr"))
                self.publics[s["name"]] = {"polType": ref["type"], "polId": ref["id"] + off, "idx": self.const_eval(s["e"]),
                                           "id": len(self.publics), "name": s["name"]}

    def scope(self, e):
        """bind every polynomial reference to the namespace it was written in"""
        if e["op"] == "pol":
            e["ns"] = self.namespace
        for v in e.get("values", []):
            self.scope(v)
        if "idxExp" in e:
            self.scope(e["idxExp"])
        return e

    def lookup(self, name, idx_exp=None, ns=None):
        ns = ns or self.namespace
        for full in ([name] if "." in name else [ns + "." + name, "Global." + name]):
            if full in self.references:
                ref = self.references[full]
                off = 0
                if idx_exp is not None:
                    off = self.const_eval(idx_exp)
                    if not ref["isArray"] or off >= ref["len"]:
                        raise PilError("bad array access to " + full)
                elif ref["isArray"]:
                    raise PilError(full + " is an array")
                return ref, off
        raise PilError("polynomial %s not defined" % name)

    def const_eval(self, e):
        op = e["op"]
        if op == "number":
            return e["value"]
        if op == "constant" or (op == "pol" and e["name"] in self.constants and not e.get("next")):
            return self.constants[e["name"]]
        if op == "neg":
            return -self.const_eval(e["values"][0])
        a, b = (self.const_eval(v) for v in e["values"])
        return {"add": a + b, "sub": a - b, "mul": a * b, "pow": a ** b if op == "pow" else 0}[op]

    # ---- pass 2: degrees, quotient reductions, folding -------------------------------------------
    def reduce_to_1(self, e):
        if e["deg"] <= 1:
            return
        if e["deg"] > 2:
            raise PilError("degree too high")
        e["idQ"] = self.n_q; self.n_q += 1
        e["deg"] = 1

    def simplify(self, e):
        if e.get("simplified"):
            return e
        op = e["op"]
        num = lambda v: {"op": "number", "deg": 0, "value": v % P, "simplified": True}
        if op == "number":
            return num(e["value"])
        if op in ("constant", "pow") or (op == "pol" and e["name"] in self.constants):
            return num(self.const_eval(e))
        if op == "public":
            if e["name"] not in self.publics:
                raise PilError("public %s not defined" % e["name"])
            e.update(deg=0, simplified=True)
            return e
        if op == "pol":
            ref, off = self.lookup(e["name"], e.get("idxExp"), e.get("ns"))
            e.update(simplified=True, ref=ref, off=off, deg=1)
            if ref["type"] == "imP":
                self.expressions[ref["id"]] = t = self.simplify(self.expressions[ref["id"]])
                self.reduce_to_1(t)
            return e
        if op == "neg":
            a = self.simplify(e["values"][0])
            if a["op"] == "number":
                return num(-a["value"])
            e.update(values=[a], deg=a["deg"], simplified=True)
            return e
        a, b = self.simplify(e["values"][0]), self.simplify(e["values"][1])
        an, bn = a["op"] == "number", b["op"] == "number"
        if op == "add":
            if an and bn: return num(a["value"] + b["value"])
            if an and a["value"] == 0: return b
            if bn and b["value"] == 0: return a
            deg = max(a["deg"], b["deg"])
        elif op == "sub":
            if an and bn: return num(a["value"] - b["value"])
            deg = max(a["deg"], b["deg"])
        elif op == "mul":
            if an and bn: return num(a["value"] * b["value"])
            if an and a["value"] == 0: return a
            if bn and b["value"] == 0: return b
            if an and a["value"] == 1: return b
            if bn and b["value"] == 1: return a
            deg = a["deg"] + b["deg"]
        else:
            raise PilError("unknown operator " + op)
        e.update(values=[a, b], deg=deg, simplified=True)
        return e

    def finish(self):
        for pi in self.pol_ids:
            self.expressions[pi["e"]] = e = self.simplify(self.expressions[pi["e"]])
            if e["deg"] > 2:
                raise PilError("%s:%d: degree too high" % (pi["fileName"], pi["line"]))
        for group in (self.plookups, self.permutations):
            for d in group:
                for k in d["f"] + d["t"] + [d["selF"], d["selT"]]:
                    if k is not None:
                        self.expressions[k] = self.simplify(self.expressions[k])
                        self.reduce_to_1(self.expressions[k])
        for d in self.connections:
            for k in d["pols"] + d["connections"]:
                self.expressions[k] = self.simplify(self.expressions[k])
                self.reduce_to_1(self.expressions[k])
        for i in range(len(self.expressions)):
            self.expressions[i] = self.simplify(self.expressions[i])

    # ---- output -----------------------------------------------------------------------------------
    def node_json(self, e, deps):
        out = {"op": e["op"], "deg": e["deg"]}
        if "idQ" in e:
            out["idQ"] = e["idQ"]
        if e["op"] == "pol":
            ref = e["ref"]
            out["op"] = {"cmP": "cm", "constP": "const", "imP": "exp"}[ref["type"]]
            out["id"], out["next"] = ref["id"] + e["off"], e["next"]
            if ref["type"] == "imP" and ref["id"] not in deps:
                deps.append(ref["id"])
        elif e["op"] == "public":
            out["id"] = self.publics[e["name"]]["id"]
        elif e["op"] == "number":
            out["value"] = str(e["value"])
        else:
            out["values"] = [self.node_json(v, deps) for v in e["values"]]
        return out

    def to_json(self):
        exprs = []
        for e in self.expressions:
            deps = []
            j = self.node_json(e, deps)
            if deps:
                j["deps"] = deps
            exprs.append(j)
        strip = lambda d, keys: {k: d[k] for k in keys}
        return {
            "nCommitments": self.n_cm, "nQ": self.n_q, "nIm": self.n_im, "nConstants": self.n_const,
            "publics": sorted(self.publics.values(), key=lambda p: p["id"]),
            "references": self.references,
            "expressions": exprs,
            "polIdentities": [strip(p, ("e", "fileName", "line")) for p in self.pol_ids],
            "plookupIdentities": [strip(p, ("f", "t", "selF", "selT", "fileName", "line")) for p in self.plookups],
            "permutationIdentities": [strip(p, ("f", "t", "selF", "selT", "fileName", "line")) for p in self.permutations],
            "connectionIdentities": [strip(p, ("pols", "connections", "fileName", "line")) for p in self.connections],
        }


def compile_pil(path, text=None, sources=None):
    """path: the main .pil (includes are resolved next to it); text overrides the main file's contents,
    sources {basename: text} those of included files"""
    c = Compiler()
    c.sources = dict(sources or {})
    c.load(path, text)
    c.finish()
    return c.to_json()


def dumps(pil):
    """the text pilcom's CLI writes: JSON.stringify(pil, null, 1)"""
    return json.dumps(pil, indent=1)


def main():
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("pil")
    ap.add_argument("-o", "--output")
    ap.add_argument("--nbits", type=int, help="replace the source's `let N: int = 2**k` by 2**nbits")
    a = ap.parse_args()
    text = open(a.pil).read()
    if a.nbits is not None:
        text, n = re.subn(r"(let\s+N\s*:\s*int\s*=|constant\s+%N\s*=)\s*2\*\*\d+", lambda m: "%s 2**%d" % (m.group(1), a.nbits), text)
        assert n == 1, "no `N = 2**k` definition to resize"
    out = dumps(compile_pil(a.pil, text))
    if a.output:
        open(a.output, "w").write(out)
    else:
        print(out)


if __name__ == "__main__":
    main()

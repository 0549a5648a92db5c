#!/usr/bin/env python3
"""Transcribe the reference's own known-answer tests for the tokenize() path into
JSON fixtures (tests/golden/*.json + tests/golden/resources/*).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The Rust reference cannot be executed here (no cargo/rustc), so these vectors are
*parsed out of the reference's test sources*, not produced by running it:

  * vibrato/src/tests/tokenizer.rs        (15 end-to-end tests, exact total_cost)
  * vibrato/src/tokenizer.rs:208-361      (4 inline-dictionary tests)
  * vibrato/src/lib.rs:8-45               (doctest)
  * vibrato/src/tests/lexicon.rs:8-78, dictionary/lexicon.rs:232-272
  * vibrato/src/tests/connector.rs:5-14, connector/matrix_connector.rs:131-183
  * vibrato/src/dictionary/character.rs:288-298
  * vibrato/src/dictionary/connector/raw_connector/scorer.rs:348-480 (Scorer), raw_connector.rs:324-510,
    dual_connector.rs:281-360 (compact connectors: builder vectors, cost(), id mapping)

The dictionary text fixtures under vibrato/src/tests/resources/ (lex.csv,
matrix.def, char.def, unk.def, user.csv; credits in resources/README.md there)
are the *inputs* of those vectors and are copied verbatim as data.
"""
import json
import os
import re
import shutil

REF = "/root/reference/vibrato/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def rust_str(lit: str) -> str:
    """Unescape the body of a Rust "..." literal."""
    out = []
    i = 0
    while i < len(lit):
        c = lit[i]
        if c == "\\":
            n = lit[i + 1]
            if n == "n":
                out.append("\n"); i += 2
            elif n == "t":
                out.append("\t"); i += 2
            elif n == '"':
                out.append('"'); i += 2
            elif n == "\\":
                out.append("\\"); i += 2
            elif n == "\n":  # line continuation: skip newline + leading whitespace
                i += 2
                while i < len(lit) and lit[i] in " \t\n":
                    i += 1
            else:
                raise ValueError("escape " + n)
        else:
            out.append(c); i += 1
    return "".join(out)


STR = r'"((?:[^"\\]|\\.)*)"'


def split_tests(src: str):
    """Yield (name, body) for each `fn test_*() { ... }`."""
    for m in re.finditer(r"fn (test_\w+)\(\) \{", src):
        depth, i = 1, m.end()
        in_str = False
        while depth:
            c = src[i]
            if in_str:
                if c == "\\":
                    i += 1
                elif c == '"':
                    in_str = False
            else:
                if c == '"':
                    in_str = True
                elif c == "{":
                    depth += 1
                elif c == "}":
                    depth -= 1
            i += 1
        yield m.group(1), src[m.end():i - 1]


def eval_cost(expr: str, costs: dict) -> int:
    expr = re.sub(r"worker\.token\((\d+)\)\.total_cost\(\)", lambda m: str(costs[int(m.group(1))]), expr)
    expr = expr.replace("\n", " ").strip()
    assert re.fullmatch(r"[-+ 0-9()]+", expr), expr
    return int(eval(expr))


def parse_sentences(body: str):
    parts = re.split(r"worker\.reset_sentence\(" + STR + r"\);", body)
    # parts = [pre, text1, seg1, text2, seg2, ...]
    sents = []
    for k in range(1, len(parts), 2):
        text = rust_str(parts[k])
        seg = parts[k + 1]
        m = re.search(r"assert_eq!\(worker\.num_tokens\(\), (\d+)\);", seg)
        sent = {"text": text, "num_tokens": int(m.group(1)), "tokens": []}
        toks = {}
        for tm in re.finditer(r"let t\d* = worker\.token\((\d+)\);(.*?)\n\s*\}", seg, re.S):
            i = int(tm.group(1)); blk = tm.group(2)
            t = {"index": i}
            s = re.search(r"t\d*\.surface\(\),\s*" + STR, blk)
            if s: t["surface"] = rust_str(s.group(1))
            s = re.search(r"t\d*\.range_char\(\), (\d+)\.\.(\d+)", blk)
            if s: t["range_char"] = [int(s.group(1)), int(s.group(2))]
            s = re.search(r"t\d*\.range_byte\(\), (\d+)\.\.(\d+)", blk)
            if s: t["range_byte"] = [int(s.group(1)), int(s.group(2))]
            s = re.search(r"t\d*\.feature\(\),\s*" + STR, blk)
            if s: t["feature"] = rust_str(s.group(1))
            s = re.search(r"t\d*\.total_cost\(\), (-?\d+)\)", blk)
            if s: t["total_cost"] = int(s.group(1))
            toks[i] = t
        # doctest style: let t0 = worker.token(0); asserts follow inline
        for tm in re.finditer(r"let (t\d+) = worker\.token\((\d+)\);(.*?)(?=let t\d+ = worker|\Z)", seg, re.S):
            i = int(tm.group(2))
            if i in toks and len(toks[i]) > 1:
                continue
            v, blk = tm.group(1), tm.group(3)
            t = {"index": i}
            s = re.search(v + r"\.surface\(\),\s*" + STR, blk)
            if s: t["surface"] = rust_str(s.group(1))
            s = re.search(v + r"\.range_char\(\), (\d+)\.\.(\d+)", blk)
            if s: t["range_char"] = [int(s.group(1)), int(s.group(2))]
            s = re.search(v + r"\.range_byte\(\), (\d+)\.\.(\d+)", blk)
            if s: t["range_byte"] = [int(s.group(1)), int(s.group(2))]
            s = re.search(v + r"\.feature\(\),\s*" + STR, blk)
            if s: t["feature"] = rust_str(s.group(1))
            toks[i] = t
        costs = {}
        for cm in re.finditer(r"assert_eq!\(\s*worker\.token\((\d+)\)\.total_cost\(\),\s*(.*?)\s*\);", seg, re.S):
            i = int(cm.group(1))
            costs[i] = eval_cost(cm.group(2), costs)
            toks.setdefault(i, {"index": i})["total_cost"] = costs[i]
        sent["tokens"] = [toks[i] for i in sorted(toks)]
        sents.append(sent)
    return sents



def split_fns(src: str):
    """Yield (name, body) for each `fn <name>_test() { ... }` of a #[cfg(test)] module."""
    src = src[src.index("#[cfg(test)]"):]
    for m in re.finditer(r"fn (\w+_test)\(\) \{", src):
        depth, i = 1, m.end()
        in_str = False
        while depth:
            c = src[i]
            if in_str:
                if c == "\\":
                    i += 1
                elif c == '"':
                    in_str = False
            else:
                if c == '"':
                    in_str = True
                elif c == "{":
                    depth += 1
                elif c == "}":
                    depth -= 1
            i += 1
        yield m.group(1), src[m.end():i - 1]


U31 = r"(?:U31::new\((\d+)\)\.unwrap\(\)|(INVALID_FEATURE_ID))"


def u31_list(text: str):
    return [0x7FFFFFFF if inv else int(num) for num, inv in re.findall(U31, text)]


def connector_vectors():
    """Scorer / RawConnector / DualConnector known-answer tests, parsed from the reference's test modules."""
    conn = os.path.join(REF, "dictionary/connector")
    out = {"scorer": [], "bigram_connector": [], "parse_cost": [], "parse_cost_errors": [], "parse_features": [], "parse_features_errors": []}
    src = open(os.path.join(conn, "raw_connector/scorer.rs"), encoding="utf-8").read()
    for name, body in split_fns(src):
        if name == "u31x8_encode_decode_test":
            continue
        ins = [[int(a), int(b), int(c)] for a, b, c in
               re.findall(r"builder\.insert\(U31::new\((\d+)\)\.unwrap\(\), U31::new\((\d+)\)\.unwrap\(\), (-?\d+)\)", body)]
        case = {"source": "vibrato/src/dictionary/connector/raw_connector/scorer.rs::" + name, "insert": ins, "retrieve": [], "accumulate": []}
        for a, b, exp in re.findall(r"scorer\.retrieve_cost\(U31::new\((\d+)\)\.unwrap\(\), U31::new\((\d+)\)\.unwrap\(\)\),\s*(Some\(-?\d+\)|None)", body):
            case["retrieve"].append([int(a), int(b), None if exp == "None" else int(exp[5:-1])])
        for k1, k2, exp in re.findall(r"scorer\.accumulate_cost\(\s*&(.*?),\s*&(.*?),?\s*\),\s*(-?\d+),?\s*\)", body, re.S):
            case["accumulate"].append([u31_list(k1), u31_list(k2), int(exp)])
        assert case["retrieve"] or case["accumulate"], name
        out["scorer"].append(case)
    for fname in ["raw_connector.rs", "dual_connector.rs"]:
        src = open(os.path.join(conn, fname), encoding="utf-8").read()
        for name, body in split_fns(src):
            if name in ("from_readers_test", "mapping_test"):
                d = {}
                for var in ["right_rdr", "left_rdr", "cost_rdr"]:
                    m = re.search(r"let " + var + r" = " + STR, body, re.S)
                    d[var] = rust_str(m.group(1))
                mp = re.search(r"ConnIdMapper::new\(vec!\[([^\]]*)\], vec!\[([^\]]*)\]\)", body)
                out["bigram_connector"].append({
                    "source": "vibrato/src/dictionary/connector/" + fname + "::" + name,
                    "dual": fname.startswith("dual"),
                    "right": d["right_rdr"], "left": d["left_rdr"], "cost": d["cost_rdr"],
                    "map": [[int(x) for x in mp.group(1).split(",")], [int(x) for x in mp.group(2).split(",")]] if mp else None,
                    "costs": [[int(r), int(l), int(c)] for r, l, c in re.findall(r"conn\.cost\((\d+), (\d+)\), (-?\d+)\)", body)],
                })
            elif name == "parse_cost_test":  # ids handed out in order of first appearance, from EMPTY maps here
                calls = re.findall(r"parse_cost\(\s*" + STR + r".*?\(U31::new\((\d+)\)\.unwrap\(\), U31::new\((\d+)\)\.unwrap\(\), (-?\d+)\)", body, re.S)
                out["parse_cost"].append({"source": "vibrato/src/dictionary/connector/raw_connector.rs::" + name,
                                          "lines": [[rust_str(l), int(r), int(lf), int(c)] for l, r, lf, c in calls]})
            elif name.startswith("parse_cost_invalid"):
                out["parse_cost_errors"].append({"source": "vibrato/src/dictionary/connector/raw_connector.rs::" + name,
                                                 "line": rust_str(re.search(r"parse_cost\(\s*" + STR, body).group(1))})
            elif name in ("parse_feature_test", "parse_feature_invalid_id_test"):
                idmap = {rust_str(k): int(v) for k, v in re.findall(STR + r"\.to_string\(\) => U31::new\((\d+)\)", body)}
                m = re.search(r"parse_features\(\s*" + STR, body)
                item = {"source": "vibrato/src/dictionary/connector/raw_connector.rs::" + name, "id_map": idmap, "line": rust_str(m.group(1))}
                if name == "parse_feature_test":
                    tail = body[m.end():]
                    item["id"] = int(re.search(r"\(\s*(\d+),\s*vec!", tail).group(1))
                    item["features"] = u31_list(tail[tail.index("vec!"):])
                    out["parse_features"].append(item)
                else:
                    out["parse_features_errors"].append(item)
    return out


def main():
    res_dst = os.path.join(HERE, "resources")
    os.makedirs(res_dst, exist_ok=True)
    for f in ["lex.csv", "matrix.def", "char.def", "unk.def", "user.csv"]:
        shutil.copyfile(os.path.join(REF, "tests/resources", f), os.path.join(res_dst, f))

    cases = []
    # 1. vibrato/src/tests/tokenizer.rs
    src = open(os.path.join(REF, "tests/tokenizer.rs"), encoding="utf-8").read()
    for name, body in split_tests(src):
        mg = re.search(r"\.max_grouping_len\((\d+)\)", body)
        cases.append({
            "name": name,
            "source": "vibrato/src/tests/tokenizer.rs",
            "dict": "fixture",
            "user": "reset_user_lexicon_from_reader(Some(USER_CSV" in body,
            "ignore_space": ".ignore_space(true)" in body,
            "max_grouping_len": int(mg.group(1)) if mg else 0,
            "sentences": parse_sentences(body),
        })
    # 2. vibrato/src/tokenizer.rs inline tests + token.rs test_iter
    for path in ["tokenizer.rs", "token.rs"]:
        src = open(os.path.join(REF, path), encoding="utf-8").read()
        for name, body in split_tests(src):
            d = {}
            for var in ["lexicon_csv", "matrix_def", "char_def", "unk_def"]:
                m = re.search(r"let " + var + r" = " + STR + ";", body)
                d[var] = rust_str(m.group(1))
            cases.append({
                "name": name,
                "source": "vibrato/src/" + path,
                "dict": {"lex": d["lexicon_csv"], "matrix": d["matrix_def"], "char": d["char_def"], "unk": d["unk_def"]},
                "user": False, "ignore_space": False, "max_grouping_len": 0,
                "sentences": parse_sentences(body),
            })
    # 3. lib.rs doctest
    src = open(os.path.join(REF, "lib.rs"), encoding="utf-8").read()
    doc = "\n".join(l[4:] if l.startswith("//! ") else l[3:] for l in src.split("\n") if l.startswith("//!"))
    doc = doc[doc.index("let tokenizer"):doc.index("# Ok(())")]
    cases.append({
        "name": "lib_doctest", "source": "vibrato/src/lib.rs:8-45", "dict": "fixture",
        "user": False, "ignore_space": False, "max_grouping_len": 0,
        "sentences": parse_sentences(doc),
    })
    n_tok = sum(len(s["tokens"]) for c in cases for s in c["sentences"])
    n_cost = sum(1 for c in cases for s in c["sentences"] for t in s["tokens"] if "total_cost" in t)
    json.dump({"cases": cases}, open(os.path.join(HERE, "tokenize_golden.json"), "w"), ensure_ascii=False, indent=1)
    print(f"tokenize_golden.json: {len(cases)} cases, {n_tok} token vectors, {n_cost} exact total_costs")

    # 4. unit-level vectors (small; transcribed by hand, each with its source)
    unit = {
        "lexicon_common_prefix": [
            {"source": "vibrato/src/tests/lexicon.rs:8-40", "dict": "fixture", "input": "東京都に行く",
             "expect": [[4, 1, 7, 7, 4675], [5, 2, 6, 6, 2816], [6, 3, 6, 8, 5320]]},
            {"source": "vibrato/src/tests/lexicon.rs:42-57", "dict": "fixture", "input": "X",
             "expect": [[i, 1, 8, 8, -20000] for i in range(40, 46)]},
            {"source": "vibrato/src/dictionary/lexicon.rs:232-272",
             "lex": "東京,1,2,3,\n東京都,4,5,6,\n東京,7,8,9,\n京都,10,11,12,\n", "input": "東京都",
             "expect": [[0, 2, 1, 2, 3], [2, 2, 7, 8, 9], [1, 3, 4, 5, 6]]},
        ],
        "word_feature": [
            {"source": "vibrato/src/tests/lexicon.rs:59-78", "word_id": 0, "feature": "た,助動詞,*,*,*,助動詞-タ,終止形-一般,タ,た,*,A,*,*,*,*"},
            {"source": "vibrato/src/tests/lexicon.rs:59-78", "word_id": 2, "feature": "に,助詞,格助詞,*,*,*,*,ニ,に,*,A,*,*,*,*"},
            {"source": "vibrato/src/tests/lexicon.rs:59-78", "word_id": 39, "feature": " ,空白,*,*,*,*,*, , ,*,A,*,*,*,*"},
            {"source": "vibrato/src/tests/lexicon.rs:59-78", "word_id": 45, "feature": "X,名詞,固有名詞,地名,一般,*,*,X,X,*,A,*,*,*,*"},
        ],
        "connector": [
            {"source": "vibrato/src/tests/connector.rs:5-14", "matrix": "fixture", "num_left": 10, "num_right": 10,
             "costs": [[0, 0, 0], [0, 1, 863], [1, 0, -3689], [9, 9, -2490]]},
            {"source": "vibrato/src/dictionary/connector/matrix_connector.rs:135-147",
             "matrix": "2 2\n0 0 0\n0 1 1\n1 0 -2\n1 1 -3", "num_left": 2, "num_right": 2,
             "costs": [[0, 0, 0], [0, 1, 1], [1, 0, -2], [1, 1, -3]]},
            {"source": "vibrato/src/dictionary/connector/matrix_connector.rs:149-165",
             "matrix": "2 3\n0 0 0\n0 1 1\n0 2 2\n1 0 -3\n1 1 -4\n1 2 -5", "num_left": 3, "num_right": 2,
             "costs": [[0, 0, 0], [0, 1, 1], [0, 2, 2], [1, 0, -3], [1, 1, -4], [1, 2, -5]]},
        ],
        "connector_errors": [
            {"source": "vibrato/src/dictionary/connector/matrix_connector.rs:185-263", "matrix": m}
            for m in ["2\n0 0 0\n0 1 1\n1 0 -2\n1 1 -3", "2 2 2\n0 0 0\n0 1 1\n1 0 -2\n1 1 -3",
                      "2 2\n0 0 0\n0 1 1\n1 -2\n1 1 -3", "2 2\n0 0 0\n0 1 1\n1 0 1 -2\n1 1 -3", "65536 65536",
                      "2 2\n0 0 0\n0 1 1\n1 2 -2\n1 1 -3", "2 2\n0 0 0\n0 1 1\n2 0 -2\n1 1 -3"]
        ],
        "char_info": [
            {"source": "vibrato/src/dictionary/character.rs:288-298", "char_def": "DEFAULT 0 1 0\nSPACE 0 1 0\n0x0020 SPACE",
             "cp": 0x20, "cate_idset": 2, "base_id": 1, "invoke": 0, "group": 1, "length": 0},
        ],
        "char_def_errors": [
            {"source": "vibrato/src/dictionary/character.rs:300-366", "char_def": c}
            for c in ["DEFAULT 0 1 0\n0x0..0xFFFF INVALID", "USER_DEFINED 0 1 0", "DEFAULT 2 1 0", "DEFAULT 0 2 0", "DEFAULT 0 2 -1",
                      "DEFAULT 0 2", "DEFAULT 0 1 0\n0x10000 DEFAULT", "DEFAULT 0 1 0\n0x0..0x10000 DEFAULT",
                      "DEFAULT 0 1 0\n0x0020..0x0019 DEFAULT"]
        ],
        "char_def_ok": [
            {"source": "vibrato/src/dictionary/character.rs:349-353", "char_def": "DEFAULT 0 1 0\n0x0..0xFFFF DEFAULT"}
        ],
    }
    unit.update(connector_vectors())
    json.dump(unit, open(os.path.join(HERE, "unit_golden.json"), "w"), ensure_ascii=False, indent=1)
    print("unit_golden.json written")


if __name__ == "__main__":
    main()

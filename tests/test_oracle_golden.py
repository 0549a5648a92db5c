"""Pin the CPU oracle against every known-answer vector the reference's own tests hold
for the tokenize() path.  CPU only."""
import pytest

from oracle import oracle as ora


def _dict_for(case, src):
    if case["dict"] == "fixture":
        d = ora.Dictionary.from_sources(src["lex.csv"], src["matrix.def"], src["char.def"], src["unk.def"])
    else:
        c = case["dict"]
        d = ora.Dictionary.from_sources(c["lex"], c["matrix"], c["char"], c["unk"])
    if case["user"]:
        d.reset_user_lexicon(src["user.csv"])
    return d


def check_case(case, worker):
    for sent in case["sentences"]:
        worker.reset_sentence(sent["text"])
        worker.tokenize()
        assert worker.num_tokens() == sent["num_tokens"], (case["name"], sent["text"])
        for exp in sent["tokens"]:
            got = worker.token(exp["index"])
            for k in ["surface", "range_char", "range_byte", "feature", "total_cost"]:
                if k in exp:
                    assert got[k] == exp[k], (case["name"], sent["text"], exp["index"], k)


def test_tokenize_golden(tokenize_golden, fixture_sources):
    n_checked = 0
    for case in tokenize_golden:
        d = _dict_for(case, fixture_sources)
        tok = ora.Tokenizer(d, case["ignore_space"], case["max_grouping_len"])
        check_case(case, tok.new_worker())
        n_checked += 1
    assert n_checked == 21


def test_lexicon_common_prefix(unit_golden, fixture_sources):
    s = fixture_sources
    for c in unit_golden["lexicon_common_prefix"]:
        if c.get("dict") == "fixture":
            d = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
        else:
            d = ora.Dictionary.from_sources(c["lex"], "12 12\n", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
        assert d.common_prefix(c["input"]) == c["expect"], c["source"]


def test_word_feature(unit_golden, fixture_sources):
    s = fixture_sources
    d = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    assert d.num_words(0) == 46
    for c in unit_golden["word_feature"]:
        assert d.word_feature(0, c["word_id"]) == c["feature"]


def test_connector(unit_golden, fixture_sources):
    for c in unit_golden["connector"]:
        m = fixture_sources["matrix.def"] if c["matrix"] == "fixture" else c["matrix"]
        d = ora.Dictionary.from_sources("a,0,0,0,x", m, "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
        assert d.num_left == c["num_left"] and d.num_right == c["num_right"]
        for r, l, cost in c["costs"]:
            assert d.conn_cost(r, l) == cost
    for c in unit_golden["connector_errors"]:
        with pytest.raises(ora.OracleError):
            ora.Dictionary.from_sources("a,0,0,0,x", c["matrix"], "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")


def test_char_def(unit_golden):
    for c in unit_golden["char_info"]:
        d = ora.Dictionary.from_sources("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")
        ci = d.char_info(c["cp"])
        for k in ["cate_idset", "base_id", "invoke", "group", "length"]:
            assert ci[k] == c[k]
    for c in unit_golden["char_def_errors"]:
        with pytest.raises(ora.OracleError):
            ora.Dictionary.from_sources("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")
    for c in unit_golden["char_def_ok"]:
        ora.Dictionary.from_sources("a,0,0,0,x", "1 1\n0 0 0", c["char_def"], "DEFAULT,0,0,0,*")


def test_ignore_space_needs_space_category():
    d = ora.Dictionary.from_sources("a,0,0,0,x", "1 1\n0 0 0", "DEFAULT 0 1 0", "DEFAULT,0,0,0,*")
    with pytest.raises(ora.OracleError):
        ora.Tokenizer(d, ignore_space=True)


def test_invalid_connection_ids(fixture_sources):
    s = fixture_sources
    with pytest.raises(ora.OracleError):
        ora.Dictionary.from_sources("a,10,0,0,x", s["matrix.def"], s["char.def"], s["unk.def"])
    d = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    with pytest.raises(ora.OracleError):
        d.reset_user_lexicon("a,0,10,0,x")


def test_output_formats(fixture_sources):
    """tokenize/src/main.rs:83-127"""
    s = fixture_sources
    d = ora.Dictionary.from_sources(s["lex.csv"], s["matrix.def"], s["char.def"], s["unk.def"])
    w = ora.Tokenizer(d).new_worker()
    w.reset_sentence("京都東京都")
    w.tokenize()
    assert ora.format_tokens(w, "wakati") == "京都 東京都\n"
    assert ora.format_tokens(w, "mecab") == (
        "京都\t京都,名詞,固有名詞,地名,一般,*,*,キョウト,京都,*,A,*,*,*,1/5\n"
        "東京都\t東京都,名詞,固有名詞,地名,一般,*,*,トウキョウト,東京都,*,B,5/9,*,5/9,*\nEOS\n")
    assert ora.format_tokens(w, "detail").split("\n")[0].endswith(
        "\tlex_type=System\tleft_id=6\tright_id=6\tword_cost=5293\ttotal_cost=5214")
    w.reset_sentence("")
    w.tokenize()
    assert ora.format_tokens(w, "mecab") == "EOS\n"

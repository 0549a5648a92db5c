"""A second, independent restatement of the reference's tokenize() path in plain Python -- TEST INFRASTRUCTURE ONLY.

Written from the Rust sources (file:line below, relative to /root/reference/vibrato/src), not from oracle/vibrato_oracle.c: no double
array (a dict of surfaces), no packed records, Python lists for the lattice.  tests/test_oracle_vs_python_restatement.py runs it against
the C oracle on random small dictionaries built to provoke ties, unknown-word rules and space handling; slow by design, small inputs only.
"""
LEX_SYSTEM, LEX_USER, LEX_UNKNOWN = 0, 1, 2  # dictionary.rs:30-40
I32_MAX = 2**31 - 1


def _wrap_i32(x):  # release-mode i32 arithmetic wraps (lattice.rs:125,139)
    return ((x + 2**31) % 2**32) - 2**31


def parse_rows(csv_text):
    """surface,left,right,cost,feature... rows without quoting (what the test generator emits); rows with an empty surface are
    skipped (lexicon.rs:179-183)."""
    rows = []
    for line in csv_text.split("\n"):
        if not line:
            continue
        surface, left, right, cost, feature = line.split(",", 4)
        if surface == "":
            continue
        rows.append((surface, int(left), int(right), int(cost), feature))
    return rows


class CharProp:
    """character.rs:140-281: categories get ids by first appearance (DEFAULT = 0), range lines are applied in file order after all
    category lines, the first category of a range line is the base category, cate_idset = OR of the listed categories' bits."""

    def __init__(self, text):
        cate_map = {"DEFAULT": 0}
        info = {}
        ranges = []
        for line in text.split("\n"):
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            cols = line.split()
            if not line.startswith("0x"):
                cid = cate_map.setdefault(cols[0], len(cate_map))
                info[cid] = (cols[1] == "1", cols[2] == "1", int(cols[3]))
            else:
                r = cols[0].split("..")
                start = int(r[0][2:], 16)
                end = int(r[1][2:], 16) + 1 if len(r) > 1 else start + 1
                cats = []
                for c in cols[1:]:
                    if c.startswith("#"):
                        break
                    cats.append(c)
                ranges.append((start, end, cats))
        self.cate_map = cate_map

        def encode(cats):
            base = cate_map[cats[0]]
            invoke, group, length = info[base]
            idset = 0
            for c in cats:
                idset |= 1 << cate_map[c]
            return {"idset": idset, "base": base, "invoke": invoke, "group": group, "length": length}

        self.table = [encode(["DEFAULT"])] * 65536
        for start, end, cats in ranges:
            ci = encode(cats)
            for cp in range(start, end):
                self.table[cp] = ci

    def char_info(self, ch):  # character.rs:112-116
        cp = ord(ch)
        return self.table[cp] if cp < 65536 else self.table[0]


class PyTokenizer:
    def __init__(self, lex_csv, num_right, num_left, matrix, char_def, unk_def, user_csv=None, ignore_space=False, max_grouping_len=0):
        self.matrix, self.num_right, self.num_left = matrix, num_right, num_left  # matrix[left * num_right + right] (matrix_connector.rs:47,82)
        self.cp = CharProp(char_def)
        self.sys = self._lexicon(parse_rows(lex_csv))
        self.user = self._lexicon(parse_rows(user_csv)) if user_csv is not None else None
        # unk.def rows grouped by category id, file order inside a category; word id = index in that order (unknown.rs:230-263)
        by_cate = [[] for _ in self.cp.cate_map]
        for name, left, right, cost, _ in parse_rows(unk_def):
            by_cate[self.cp.cate_map[name]].append((left, right, cost))
        self.unk_offsets, self.unk_entries = [], []
        for v in by_cate:
            self.unk_offsets.append(len(self.unk_entries))
            self.unk_entries.extend(v)
        self.unk_offsets.append(len(self.unk_entries))
        self.space_cateset = (1 << self.cp.cate_map["SPACE"]) if ignore_space else None  # tokenizer.rs:40-53
        self.max_grouping_len = max_grouping_len if max_grouping_len else None  # 0 = unlimited (tokenizer.rs:67-74)
        # event counters of SURVEY.md 8(d) (the algorithmic bytes of the roofline are made of these), summed over tokenize() calls
        self.counters = dict.fromkeys(["n_sentences", "n_bytes", "n_chars", "n_trie_steps", "n_trie_hits", "n_lex_matches", "n_unk_nodes",
                                       "n_nodes", "n_pairs_ref", "n_pairs_dedup", "n_tokens"], 0)

    @staticmethod
    def _lexicon(rows):
        by_surface, params = {}, []
        for wid, (surface, left, right, cost, _) in enumerate(rows):  # word id = index over kept rows
            by_surface.setdefault(surface, []).append(wid)
            params.append((left, right, cost))
        return {"by_surface": by_surface, "params": params}

    def _prefixes(self, lex, chars, start):
        """lexicon.rs:33-46: matches in increasing end_char; one surface's entries in ascending word id.  Counts what a trie walk from
        `start` does: one step per character tried -- the last one failing unless the sentence ends first -- and one hit per surface found."""
        if "prefixes" not in lex:
            lex["prefixes"] = {s[:k] for s in lex["by_surface"] for k in range(1, len(s) + 1)}
        c = self.counters
        for length in range(1, len(chars) - start + 1):
            c["n_trie_steps"] += 1
            key = "".join(chars[start:start + length])
            if key not in lex["prefixes"]:
                break
            ids = lex["by_surface"].get(key, ())
            if ids:
                c["n_trie_hits"] += 1
                c["n_lex_matches"] += len(ids)
            for wid in ids:
                yield length, wid, lex["params"][wid]

    def cost(self, right_id, left_id):
        return self.matrix[left_id * self.num_right + right_id]

    def tokenize(self, text):
        chars = list(text)
        n = len(chars)
        cnt = self.counters
        if n == 0:  # worker.rs:50-52: nothing is computed (and nothing counted) for an empty sentence
            return []
        cnt["n_sentences"] += 1
        cnt["n_bytes"] += len(text.encode("utf-8"))
        cnt["n_chars"] += n
        c2b, b = [], 0
        for ch in chars:
            c2b.append(b)
            b += len(ch.encode("utf-8"))
        c2b.append(b)
        cinfo = [self.cp.char_info(ch) for ch in chars]
        groupable = [1] * n  # sentence.rs:57-71
        for i in range(n - 1, 0, -1):
            if cinfo[i - 1]["idset"] & cinfo[i]["idset"]:
                groupable[i - 1] = groupable[i] + 1

        ends = [[] for _ in range(n + 1)]
        ends[0].append({"right": 0, "min_cost": 0, "start_node": None})  # BOS (lattice.rs:72-83)

        seen_left = set()  # left ids of the current step: search_min_node's result depends only on (start_node, left_id)

        def insert_node(start_node, start_word, end_word, lex, wid, param):  # lattice.rs:103-151
            left, right, wcost = param
            cnt["n_nodes"] += 1
            cnt["n_pairs_ref"] += len(ends[start_node])
            if left not in seen_left:
                seen_left.add(left)
                cnt["n_pairs_dedup"] += len(ends[start_node])
            min_idx, min_cost = 0xFFFF, I32_MAX
            for i, prev in enumerate(ends[start_node]):
                c = _wrap_i32(prev["min_cost"] + self.cost(prev["right"], left))
                if c <= min_cost:  # ties: the last one wins
                    min_idx, min_cost = i, c
            ends[end_word].append({"start_node": start_node, "start_word": start_word, "lex": lex, "wid": wid, "left": left, "right": right,
                                   "min_idx": min_idx, "min_cost": _wrap_i32(min_cost + wcost)})

        def unk_words(start, has_matched, emit):  # unknown.rs:69-137
            ci = cinfo[start]
            if has_matched and not ci["invoke"]:
                return

            def scan(end):
                for wid in range(self.unk_offsets[ci["base"]], self.unk_offsets[ci["base"] + 1]):
                    cnt["n_unk_nodes"] += 1
                    emit(start, end, wid, self.unk_entries[wid])

            grouped = False
            g = groupable[start]
            if ci["group"]:
                grouped = True
                if self.max_grouping_len is None or g - 1 <= self.max_grouping_len:
                    scan(start + g)
                    has_matched = True
            for i in range(1, min(ci["length"], g) + 1):
                if grouped and i == g:
                    continue
                if n < start + i:
                    break
                scan(start + i)
                has_matched = True
            if not has_matched:
                scan(start + 1)

        start_node = start_word = 0  # tokenizer.rs:94-139
        while start_word < n:
            if not ends[start_node]:
                start_word += 1
                start_node = start_word
                continue
            if self.space_cateset is not None and cinfo[start_node]["idset"] & self.space_cateset:
                start_word += groupable[start_node]
            if start_word == n:
                break
            has_matched = False  # tokenizer.rs:141-199
            seen_left.clear()
            if self.user is not None:
                for length, wid, param in self._prefixes(self.user, chars, start_word):
                    insert_node(start_node, start_word, start_word + length, LEX_USER, wid, param)
                    has_matched = True
            for length, wid, param in self._prefixes(self.sys, chars, start_word):
                insert_node(start_node, start_word, start_word + length, LEX_SYSTEM, wid, param)
                has_matched = True
            unk_words(start_word, has_matched, lambda s, e, wid, param, sn=start_node: insert_node(sn, s, e, LEX_UNKNOWN, wid, param))
            start_word += 1
            start_node = start_word

        # EOS (lattice.rs:85-101) and the back-trace (lattice.rs:159-168)
        min_idx, min_cost = 0xFFFF, I32_MAX
        for i, prev in enumerate(ends[start_node]):
            c = _wrap_i32(prev["min_cost"] + self.cost(prev["right"], 0))
            if c <= min_cost:
                min_idx, min_cost = i, c
        cnt["n_pairs_ref"] += len(ends[start_node])
        cnt["n_pairs_dedup"] += len(ends[start_node])
        tokens = []
        end, idx = start_node, min_idx
        while end != 0:
            node = ends[end][idx]
            tokens.append({"start_char": node["start_word"], "end_char": end, "start_byte": c2b[node["start_word"]], "end_byte": c2b[end],
                           "word_idx": (node["lex"] << 30) | node["wid"], "total_cost": node["min_cost"]})
            end, idx = node["start_node"], node["min_idx"]
        tokens.reverse()
        cnt["n_tokens"] += len(tokens)
        return tokens

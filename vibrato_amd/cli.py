"""`tokenize`-compatible command line (reference: tokenize/src/main.rs:31-132).

    python -m vibrato_amd.cli -i system.dic.zst [-u user.csv] [-O mecab|wakati|detail] [-S] [-M N] < lines.txt

-i takes what the reference's CLI takes -- a dictionary written by Dictionary::write, zstd-compressed or not
(tokenize/src/main.rs:34,59-60) -- or a directory of MeCab-format sources (lex.csv, matrix.def, char.def, unk.def).
--compile-to PATH writes the loaded dictionary back out as system.dic.zst (compile/src/main.rs:98).  Lines are read with the semantics of BufRead::lines()
(`\\n` / `\\r\\n` stripped) and tokenized in blocks of --block lines per GPU batch; the bytes
written are exactly those of the reference CLI for the same mode.
"""
import argparse
import os
import sys

from . import api


def read_sources(d):
    out = []
    for f in ("lex.csv", "matrix.def", "char.def", "unk.def"):
        with open(os.path.join(d, f), "rb") as fh:
            out.append(fh.read())
    return out


def main(argv=None, stdin=None, stdout=None):
    ap = argparse.ArgumentParser(prog="tokenize", description="Predicts morphemes")
    ap.add_argument("-i", "--sysdic", required=True, help="System dictionary (in zstd), or a directory with lex.csv, matrix.def, char.def, unk.def")
    ap.add_argument("--compile-to", default=None, help="also write the dictionary as system.dic.zst (zstd level 19) to this path")
    ap.add_argument("-u", "--userlex-csv", default=None, help="User lexicon file.")
    ap.add_argument("-O", "--output-mode", default="mecab", choices=["mecab", "wakati", "detail"])
    ap.add_argument("-S", "--ignore-space", action="store_true", help="Ignores white spaces in input strings.")
    ap.add_argument("-M", "--max-grouping-len", type=int, default=None, help="Maximum length of unknown words.")
    ap.add_argument("--block", type=int, default=65536, help="lines per GPU batch")
    ap.add_argument("--device", type=int, default=-1)
    args = ap.parse_args(argv)
    stdin = stdin or sys.stdin.buffer
    stdout = stdout or sys.stdout.buffer

    print("Loading the dictionary...", file=sys.stderr)
    if os.path.isdir(args.sysdic):
        d = api.SystemDictionaryBuilder.from_readers(*read_sources(args.sysdic))
    else:
        d = api.Dictionary.read(args.sysdic)
    if args.compile_to:
        with open(args.compile_to, "wb") as fh:
            d.write(fh, zstd_level=19)
    if args.userlex_csv:
        with open(args.userlex_csv, "rb") as fh:
            d.reset_user_lexicon_from_reader(fh.read())
    tok = api.Tokenizer(d, device=args.device).ignore_space(args.ignore_space).max_grouping_len(args.max_grouping_len or 0)
    print("Ready to tokenize", file=sys.stderr)

    # Three stages, each on its own thread (the library calls release the GIL): this thread reads lines and cuts blocks, one thread
    # pushes block k + 1 through the GPU (vbt_tokenize_batch), one renders block k (vbt_batch_format) and writes it -- the formatter
    # of one block runs under the kernels and copies of the next.  Output order = input order (one block at a time per stage).
    import queue
    import threading
    blocks, batches = queue.Queue(maxsize=2), queue.Queue(maxsize=2)
    failure = []

    def tokenize_stage():
        try:
            while True:
                lines = blocks.get()
                if lines is None:
                    break
                if not failure:
                    batches.put(tok.tokenize_batch(lines))
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            failure.append(e)
            while blocks.get() is not None:  # (keep draining: the reader must not block on a full queue)
                pass
        batches.put(None)

    def output_stage():
        try:
            while True:
                b = batches.get()
                if b is None:
                    break
                if not failure:
                    b.format_into(stdout.write, args.output_mode)
        except BaseException as e:  # noqa: BLE001
            failure.append(e)
            while batches.get() is not None:
                pass

    # (daemon threads, and the sentinel is posted whatever happens to the reader -- an I/O error on stdin, a KeyboardInterrupt: the stages
    # then finish what is queued and the reader's exception is raised behind the join instead of an interpreter that hangs at exit)
    stages = [threading.Thread(target=tokenize_stage, daemon=True), threading.Thread(target=output_stage, daemon=True)]
    for t in stages:
        t.start()
    reader_error = None
    try:
        block = []
        for raw in stdin:
            if raw.endswith(b"\n"):
                raw = raw[:-1]
                if raw.endswith(b"\r"):
                    raw = raw[:-1]
            block.append(raw)
            if len(block) >= args.block:
                blocks.put(block)
                block = []
        if block:
            blocks.put(block)
    except BaseException as e:  # noqa: BLE001 -- re-raised below, behind the join
        reader_error = e
        failure.append(e)  # (the stages skip what is still queued)
    finally:
        blocks.put(None)
        for t in stages:
            t.join()
    if reader_error is not None:
        raise reader_error
    if failure:
        raise failure[0]
    stdout.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())

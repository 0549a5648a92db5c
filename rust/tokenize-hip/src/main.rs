//! `tokenize` on the batched HIP path: the flags and the output bytes of the reference's command
//! (`tokenize/src/main.rs:31-132`: `-i/--sysdic`, `-u/--userlex-csv`, `-O/--output-mode mecab|wakati|detail`,
//! `-S/--ignore-space`, `-M/--max-grouping-len N`), plus `--block LINES` = lines per device batch.
//!
//! Standard input is read line by line (`\n` / `\r\n` stripped, as `BufRead::lines` does), collected into blocks and
//! handed to `Tokenizer::tokenize_batch`; `Batch::format` renders exactly what the reference prints per sentence.
//! Argument parsing is by hand so that the crate has no dependency to fetch.
use std::error::Error;
use std::fs::File;
use std::io::{self, BufRead, BufWriter, Write};
use std::process::exit;

use vibrato::batch::OutputMode;
use vibrato::{Dictionary, Tokenizer};

struct Options {
    sysdic: String,
    userlex_csv: Option<String>,
    mode: OutputMode,
    ignore_space: bool,
    max_grouping_len: usize,
    block: usize,
}

fn usage() -> ! {
    eprintln!("Predicts morphemes\n\nUsage: tokenize -i <SYSDIC> [-u <USERLEX_CSV>] [-O mecab|wakati|detail] [-S] [-M <N>] [--block <LINES>]");
    exit(2)
}

fn parse_args() -> Options {
    let mut o = Options { sysdic: String::new(), userlex_csv: None, mode: OutputMode::Mecab, ignore_space: false, max_grouping_len: 0, block: 1 << 16 };
    let mut args = std::env::args().skip(1);
    while let Some(flag) = args.next() {
        let mut value = |what: &str| args.next().unwrap_or_else(|| { eprintln!("{what} needs a value"); usage() });
        match flag.as_str() {
            "-i" | "--sysdic" => o.sysdic = value("--sysdic"),
            "-u" | "--userlex-csv" => o.userlex_csv = Some(value("--userlex-csv")),
            "-O" | "--output-mode" => {
                o.mode = match value("--output-mode").as_str() {
                    "mecab" => OutputMode::Mecab,
                    "wakati" => OutputMode::Wakati,
                    "detail" => OutputMode::Detail,
                    _ => { eprintln!("Could not parse a mode"); usage() }
                }
            }
            "-S" | "--ignore-space" => o.ignore_space = true,
            "-M" | "--max-grouping-len" => o.max_grouping_len = value("--max-grouping-len").parse().unwrap_or_else(|_| usage()),
            "--block" => o.block = value("--block").parse::<usize>().unwrap_or_else(|_| usage()).max(1),
            _ => usage(),
        }
    }
    if o.sysdic.is_empty() {
        usage()
    }
    o
}

fn run(o: Options) -> Result<(), Box<dyn Error>> {
    eprintln!("Loading the dictionary...");
    // `Dictionary::read` unwraps the zstd frame of a `system.dic.zst` itself
    let mut dict = Dictionary::read(File::open(&o.sysdic)?)?;
    if let Some(path) = &o.userlex_csv {
        dict = dict.reset_user_lexicon_from_reader(Some(File::open(path)?))?;
    }
    let tokenizer = Tokenizer::new(dict).ignore_space(o.ignore_space)?.max_grouping_len(o.max_grouping_len);
    tokenizer.build_device_image()?; // "no MI355X" is an error message here, not a panic in new_worker()
    eprintln!("Ready to tokenize");

    // Three stages (as vibrato_amd/cli.py): this thread reads lines and cuts blocks, one thread pushes block k + 1 through the GPU
    // (vbt_tokenize_batch), one renders block k (vbt_batch_format) and writes it -- the formatter of one block runs under the kernels
    // and copies of the next.  Bounded channels of depth 2 keep the order and the memory.
    let (block_tx, block_rx) = std::sync::mpsc::sync_channel::<Vec<String>>(2);
    let (batch_tx, batch_rx) = std::sync::mpsc::sync_channel(2);
    let mode = o.mode;
    let result: Result<(), Box<dyn std::error::Error + Send + Sync>> = std::thread::scope(|sc| {
        let tok = &tokenizer;
        let tokenize = sc.spawn(move || -> Result<(), Box<dyn std::error::Error + Send + Sync>> {
            for block in block_rx {
                if batch_tx.send(tok.tokenize_batch(&block)?).is_err() {
                    break; // the output stage is gone: its error is the one to report
                }
            }
            Ok(())
        });
        let output = sc.spawn(move || -> Result<(), Box<dyn std::error::Error + Send + Sync>> {
            let stdout = io::stdout();
            let mut out = BufWriter::new(stdout.lock());
            for batch in batch_rx {
                out.write_all(batch.format(mode)?.as_bytes())?;
            }
            out.flush()?;
            Ok(())
        });
        let mut block: Vec<String> = Vec::with_capacity(o.block);
        let mut read_err = None;
        for line in io::stdin().lock().lines() {
            match line {
                Ok(l) => block.push(l),
                Err(e) => { read_err = Some(e); break; }
            }
            if block.len() == o.block && block_tx.send(std::mem::replace(&mut block, Vec::with_capacity(o.block))).is_err() {
                break;
            }
        }
        if read_err.is_none() && !block.is_empty() {
            let _ = block_tx.send(block);
        }
        drop(block_tx);
        let a = tokenize.join().expect("tokenize stage panicked");
        let b = output.join().expect("output stage panicked");
        if let Some(e) = read_err { return Err(e.into()); }
        a.and(b)
    });
    result.map_err(|e| -> Box<dyn std::error::Error> { e })?;
    Ok(())
}

fn main() {
    if let Err(e) = run(parse_args()) {
        eprintln!("tokenize: {e}");
        exit(1)
    }
}

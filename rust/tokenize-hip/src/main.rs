//! `tokenize` (reference: `tokenize/src/main.rs:31-132`) on the batched HIP path: same flags, same output bytes.
//! Lines are read with `BufRead::lines()` and tokenized in blocks of `--block` lines per device batch.
use std::error::Error;
use std::fs::File;
use std::io::{BufRead, BufWriter, Write};
use std::path::PathBuf;
use std::str::FromStr;

use clap::Parser;
use vibrato::batch::OutputMode;
use vibrato::{Dictionary, Tokenizer};

#[derive(Clone, Debug)]
struct Mode(OutputMode);

impl FromStr for Mode {
    type Err = &'static str;
    fn from_str(mode: &str) -> Result<Self, Self::Err> {
        match mode {
            "mecab" => Ok(Self(OutputMode::Mecab)),
            "wakati" => Ok(Self(OutputMode::Wakati)),
            "detail" => Ok(Self(OutputMode::Detail)),
            _ => Err("Could not parse a mode"),
        }
    }
}

#[derive(Parser, Debug)]
#[clap(name = "tokenize", about = "Predicts morphemes")]
struct Args {
    /// System dictionary (in zstd).
    #[clap(short = 'i', long)]
    sysdic: PathBuf,

    /// User lexicon file.
    #[clap(short = 'u', long)]
    userlex_csv: Option<PathBuf>,

    /// Output mode. Choices are mecab, wakati, and detail.
    #[clap(short = 'O', long, default_value = "mecab")]
    output_mode: Mode,

    /// Ignores white spaces in input strings.
    #[clap(short = 'S', long)]
    ignore_space: bool,

    /// Maximum length of unknown words.
    #[clap(short = 'M', long)]
    max_grouping_len: Option<usize>,

    /// Lines per device batch.
    #[clap(long, default_value = "65536")]
    block: usize,
}

fn main() -> Result<(), Box<dyn Error>> {
    let args = Args::parse();

    eprintln!("Loading the dictionary...");
    let mut dict = Dictionary::read(File::open(args.sysdic)?)?; // the zstd frame is unwrapped by the library
    if let Some(userlex_csv) = args.userlex_csv {
        dict = dict.reset_user_lexicon_from_reader(Some(File::open(userlex_csv)?))?;
    }
    let tokenizer = Tokenizer::new(dict).ignore_space(args.ignore_space)?.max_grouping_len(args.max_grouping_len.unwrap_or(0));
    tokenizer.build_device_image()?;

    eprintln!("Ready to tokenize");
    let out = std::io::stdout();
    let mut out = BufWriter::new(out.lock());
    let mut block: Vec<String> = Vec::with_capacity(args.block);
    let mut flush = |block: &mut Vec<String>| -> Result<(), Box<dyn Error>> {
        if !block.is_empty() {
            out.write_all(tokenizer.tokenize_batch(block.iter())?.format(args.output_mode.0)?.as_bytes())?;
            block.clear();
        }
        Ok(())
    };
    for line in std::io::stdin().lock().lines() {
        block.push(line?);
        if block.len() >= args.block {
            flush(&mut block)?;
        }
    }
    flush(&mut block)?;
    Ok(())
}

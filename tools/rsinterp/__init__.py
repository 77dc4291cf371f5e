"""A small interpreter for the subset of Rust the reference's DSP code is written in.

Purpose ("pin the oracle to the reference text"): tools/rs2fixtures.py loads the reference's own source
files from /root/reference, EXECUTES the functions of the hot path (Fft, Imdct, aac Dsp::synth, the MP3 hybrid synthesis
and polyphase filterbank, Vorbis synth / floor / coupling, the FLAC and ALAC predictors, ...) on seeded inputs with
Rust's arithmetic (one IEEE rounding per f32 / f64 operation, wrapping integers, glibc libm for the table formulas)
and commits the results as bit patterns under tests/golden/.  No function body is restated by hand: what runs is the
reference's text.  This is development / test infrastructure; nothing in the product imports it.

    parser.py   lexer + recursive-descent parser (items, expressions, patterns, types, macro_rules)
    interp.py   tree-walking evaluator, value model, built-in methods of the std types the code uses
    prelude.rs  the few third-party items the reference imports (num-complex's Complex, restated from its
                published source; std::num::Wrapping is built in)
"""
from .interp import Interp, RustPanic  # noqa: F401

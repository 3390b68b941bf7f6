// test-only: /root/reference/include/moshi/moshi.h includes <sentencepiece_processor.h> (SentencePiece v0.2.0, absent from this image) for its tokenizer API;
// the LM headers compiled by tests/ref_link/ref_lm.cpp need moshi.h only for the plain struct `Entry`. Nothing of SentencePiece is declared or used.
#pragma once

"""Empty stand-in for gensim.utils (imported, never used, by the reference generator)."""

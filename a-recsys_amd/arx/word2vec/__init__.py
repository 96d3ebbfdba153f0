"""word2vec-style recommenders (reference word2vec/*): skip-gram and CBOW input models on the
same lookup / scorer / loss / sparse-Adagrad kernels as the HMF model."""

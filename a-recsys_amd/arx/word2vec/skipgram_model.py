"""Skip-gram recommender (word2vec/skipgram_model.py:12-137): the training input is
mean([user, first context item]); evaluation / recommendation average all context items."""
from .linear_seq import LinearSeq


class Model(LinearSeq):
    def __init__(self, *args, **kwargs):
        kwargs['cbow'] = False
        super().__init__(*args, **kwargs)

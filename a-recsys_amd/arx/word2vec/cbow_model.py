"""CBOW recommender (word2vec/cbow_model.py:12-140): the input is mean([user, mean of the
n context items]) for training and evaluation alike."""
from .linear_seq import LinearSeq


class Model(LinearSeq):
    def __init__(self, *args, **kwargs):
        kwargs['cbow'] = True
        super().__init__(*args, **kwargs)

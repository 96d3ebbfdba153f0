"""Attributes -- the per-entity attribute container the hot path consumes.

Same constructor / fields / setters as the reference's plain-data class
(attributes/attribute.py:7-46) so that objects built by the reference's
preprocessing (attributes/comb_attribute.py) or by arx.utils.synthetic drop in.
Layout (utils/preprocess.py:169-238):
  features_cat[f]    int[N+1]   vocabulary row of entity n for categorical f
                                (last entry = _START row)
  features_mulhot[f] int[sum len + 1]  CSR values (token rows)
  mulhot_starts[f]   int[N+2]   CSR row pointer (starts[n] = first token of n)
  mulhot_lengths[f]  int[N+1]   tokens per entity (>= 1)
and the logit-ordered copies for the full scorer (preprocess.py:240-326).
"""
from __future__ import annotations


class Attributes(object):
    def __init__(self, num_feature_cat=0, feature_cat=None, num_text_feat=0,
                 feature_mulhot=None, mulhot_max_length=None, mulhot_starts=None,
                 mulhot_lengths=None, v_sizes_cat=None, v_sizes_mulhot=None,
                 embedding_size_list_cat=None):
        self.num_features_cat = num_feature_cat
        self.num_features_mulhot = num_text_feat
        self.features_cat = feature_cat if feature_cat is not None else []
        self.features_mulhot = feature_mulhot if feature_mulhot is not None else []
        self.mulhot_max_length = mulhot_max_length
        self.mulhot_starts = mulhot_starts if mulhot_starts is not None else []
        self.mulhot_lengths = mulhot_lengths if mulhot_lengths is not None else []
        self._embedding_classes_list_cat = v_sizes_cat if v_sizes_cat is not None else []
        self._embedding_classes_list_mulhot = v_sizes_mulhot if v_sizes_mulhot is not None else []
        self._embedding_size_list_cat = embedding_size_list_cat or []
        self._embedding_size_list_mulhot = []
        self.full_cat_tr = []
        self.full_values_tr = []
        self.full_segids_tr = []
        self.full_lengths_tr = []

    def set_model_size(self, sizes, opt=0):
        """attribute.py:24-38 -- int => every feature; list => cat (opt=0) / mulhot."""
        if isinstance(sizes, list):
            if opt == 0:
                if len(sizes) != self.num_features_cat:
                    raise ValueError("need one size per categorical feature")
                self._embedding_size_list_cat = sizes
            else:
                if len(sizes) != self.num_features_mulhot:
                    raise ValueError("need one size per multi-hot feature")
                self._embedding_size_list_mulhot = sizes
        elif isinstance(sizes, int):
            self._embedding_size_list_cat = [sizes] * self.num_features_cat
            self._embedding_size_list_mulhot = [sizes] * self.num_features_mulhot
        else:
            raise ValueError("error: sizes need to be list or int")

    def set_target_prediction(self, features_cat_tr, full_values_tr, full_segids_tr,
                              full_lengths_tr):
        """attribute.py:40-47 -- logit-ordered maps for the full-vocabulary scorer."""
        self.full_cat_tr = features_cat_tr
        self.full_values_tr = full_values_tr
        self.full_segids_tr = full_segids_tr
        self.full_lengths_tr = full_lengths_tr

    def overview(self, out=None):
        p = out if out else print
        p('# of categorical attributes: {}'.format(self.num_features_cat))
        p('# of multi-hot   attributes: {}'.format(self.num_features_mulhot))

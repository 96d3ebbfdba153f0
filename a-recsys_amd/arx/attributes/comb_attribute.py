"""HET / MIX attribute combination -- py3 mirror of attributes/comb_attribute.py (same classes,
constructor arguments and methods).

HET keeps one embedding table per attribute column (categorical columns -> one-hot lookups,
multi-hot columns -> bags); MIX merges every attribute of an entity into ONE bag of
name-prefixed tokens over a single shared table.  Both end in the item <-> logit index maps:
  HET (comb_attribute.py:162-176): the items whose id token survived the vocabulary cut, in
      ascending item order;
  MIX (comb_attribute.py:83-98):   the logits_size_tr most frequent training items, most
      frequent first (ties: first appearance in the training log; the reference's order among
      ties is Python-2 dict order).
"""
from __future__ import annotations

import numpy as np
import pandas as pd

from . import attribute
from ..utils.preprocess import (create_dictionary, create_dictionary_mix, filter_cat, filter_mulhot,
                                tokenize_attribute_map)


class Comb_Attributes(object):
    def __init__(self):
        return

    def get_attributes(self, users, items, data_tr, user_features, item_features):
        """comb_attribute.py:10-68 -> (u_attributes, i_attributes, item_ind2logit_ind,
        logit_ind2item_ind): vocabularies from the training interactions, attribute maps for
        every entity, then the logit-ordered copies for the full scorer."""
        user_feature_names, user_feature_types = user_features
        item_feature_names, item_feature_types = item_features
        tr = np.asarray([(p[0], p[1]) for p in data_tr], dtype=np.int64).reshape(-1, 2)
        u_inds, i_inds_tr = tr[:, 0], tr[:, 1]

        self.create_dictionary(self.data_dir, u_inds, users, user_feature_types, user_feature_names,
                               self.max_vocabulary_size, self.logits_size_tr, prefix='user',
                               threshold=self.threshold)
        u_attributes = attribute.Attributes(*tokenize_attribute_map(
            self.data_dir, users, user_feature_types, self.max_vocabulary_size, self.logits_size_tr,
            prefix='user'))

        self.create_dictionary(self.data_dir, i_inds_tr, items, item_feature_types, item_feature_names,
                               self.max_vocabulary_size, self.logits_size_tr, prefix='item',
                               threshold=self.threshold)
        i_maps = tokenize_attribute_map(self.data_dir, items, item_feature_types,
                                        self.max_vocabulary_size, self.logits_size_tr, prefix='item')
        num_cat, features_cat = i_maps[0], i_maps[1]
        item2fea0 = features_cat[0] if len(features_cat) > 0 else None
        item_ind2logit_ind, logit_ind2item_ind = self.index_mapping(item2fea0, i_inds_tr, len(items))
        i_attributes = attribute.Attributes(*i_maps)

        features_cat_tr = filter_cat(num_cat, features_cat, logit_ind2item_ind)
        (_, full_values_tr, _, _, full_segids_tr, full_lengths_tr) = filter_mulhot(
            self.data_dir, items, item_feature_types, self.max_vocabulary_size, logit_ind2item_ind,
            prefix='item')
        i_attributes.set_target_prediction(features_cat_tr, full_values_tr, full_segids_tr,
                                           full_lengths_tr)
        return u_attributes, i_attributes, item_ind2logit_ind, logit_ind2item_ind


def _maps(order):
    order = [int(x) for x in order]
    return {e: k for k, e in enumerate(order)}, {k: e for k, e in enumerate(order)}


class MIX(Comb_Attributes):
    def __init__(self, data_dir, max_vocabulary_size=500000, logits_size_tr=50000, threshold=2):
        self.data_dir = data_dir
        self.max_vocabulary_size = max_vocabulary_size
        self.logits_size_tr = logits_size_tr
        self.threshold = threshold
        self.create_dictionary = create_dictionary_mix

    def index_mapping(self, item2fea0, i_inds, M=None):
        codes, uniq = pd.factorize(np.asarray(i_inds, dtype=np.int64), sort=False)
        cnt = np.bincount(codes, minlength=len(uniq))
        if self.logits_size_tr > len(uniq):
            raise AssertionError('Item_vocab_size should be smaller than # of appeared items')
        order = np.asarray(uniq)[np.argsort(-cnt, kind='stable')][:self.logits_size_tr]
        return _maps(order)

    @staticmethod
    def _bag_column(values, names, types):
        n = len(values)
        parts = []
        for j, (t, name) in enumerate(zip(types, names)):
            if t > 1:
                continue
            s = pd.Series(values[:, j], dtype=object).map(str)
            if t == 1:
                s = s.str.replace(',', ',' + name, regex=False)
            parts.append(name + s)
        out = np.zeros((n, 1), dtype=object)
        if parts:
            bag = parts[0]
            for p in parts[1:]:
                bag = bag + ',' + p
            out[:, 0] = bag.to_numpy(dtype=object)
        else:
            out[:, 0] = ''
        return out

    def mix_attr(self, users, items, user_features, item_features):
        """comb_attribute.py:100-148: every entity becomes one comma-joined bag
        `<column name><value>` (multi-hot columns contribute one token per element); the user id
        column is renamed 'uid'.  A table with a single categorical column stays categorical."""
        user_feature_names, user_feature_types = user_features
        item_feature_names, item_feature_types = item_features
        user_feature_names = list(user_feature_names)
        user_feature_names[0] = 'uid'
        users2 = self._bag_column(np.asarray(users, dtype=object), user_feature_names, user_feature_types)
        items2 = self._bag_column(np.asarray(items, dtype=object), item_feature_names, item_feature_types)

        def kind(types):
            return 0 if (len(types) == 1 and types[0] == 0) else 1
        return (users2, items2, (['mix'], [kind(user_feature_types)]),
                (['mix'], [kind(item_feature_types)]))


class HET(Comb_Attributes):
    def __init__(self, data_dir, max_vocabulary_size=50000, logits_size_tr=50000, threshold=2):
        self.data_dir = data_dir
        self.max_vocabulary_size = max_vocabulary_size
        self.logits_size_tr = logits_size_tr
        self.threshold = threshold
        self.create_dictionary = create_dictionary

    def index_mapping(self, item2fea0, i_inds, M):
        known = np.flatnonzero(np.asarray(item2fea0[:M]) != 0)
        if len(known) != self.logits_size_tr:
            raise AssertionError(
                'Item_vocab_size %d too large! need to be no greater than %d\nFix: --item_vocab_size '
                '[smaller item_vocab_size]\n' % (self.logits_size_tr, len(known)))
        return _maps(known)

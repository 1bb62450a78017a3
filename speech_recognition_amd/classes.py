"""Label vocabularies (reference classes.py:5-41).  The word ORDER defines the class indices
(silence 0, unknown 1, then these lists; reference input_data.py:210-212), so it is reproduced."""
from collections import OrderedDict

from .input_data import prepare_words_list

_WANTED = 'stop down off right up go on yes left no'
_ALL = ('sheila nine stop bed four six down bird marvin cat off right seven eight up three happy go zero on '
        'wow dog yes five one tree house two left no')
_REVERSED = ('new_owt new_yppah new_xis new_esuoh new_neves new_thgie new_ruof new_tac new_nivram new_enin '
             'new_aliehs new_eert new_orez new_eerht new_evif new_deb new_drib')


def get_classes(wanted_only=False, extend_reversed=False):
    classes = (_WANTED if wanted_only else _ALL).split(' ')
    assert len(classes) == (10 if wanted_only else 30)
    if extend_reversed:
        assert not wanted_only
        extra = _REVERSED.split(' ')
        assert len(extra) == 17
        classes.extend(extra)
    return classes


def get_int2label(wanted_only=False, extend_reversed=False):
    words = prepare_words_list(get_classes(wanted_only=wanted_only, extend_reversed=extend_reversed))
    return OrderedDict((i, w) for i, w in enumerate(words))


def get_label2int(wanted_only=False, extend_reversed=False):
    words = prepare_words_list(get_classes(wanted_only=wanted_only, extend_reversed=extend_reversed))
    return OrderedDict((w, i) for i, w in enumerate(words))

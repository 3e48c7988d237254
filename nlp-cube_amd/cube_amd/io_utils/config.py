"""Hyper-parameter objects that size every tensor on the parser/tagger path.

Host-side mirror of the reference's INI-backed configs (cube/io_utils/config.py):
`Config.load/save` :28-78 (one `configparser` section named after the class,
values auto-cast with `ast.literal_eval`), `TaggerConfig` :104-124,
`ParserConfig` :127-156.  File format and defaults are identical, so a released
`*-trf-parser.config` loads unchanged.
"""
import ast
import collections
import configparser
import sys


class Config(object):
    def __init__(self):
        self.__config__ = self.__class__.__name__

    @staticmethod
    def _auto_cast(s):
        try:
            return ast.literal_eval(s)
        except Exception:
            return s

    def save(self, filename):
        cp = configparser.ConfigParser()
        cp.add_section(self.__config__)
        for k, v in collections.OrderedDict(sorted(self.__dict__.items())).items():
            if k.startswith("_"):
                continue
            if isinstance(v, float):
                s = str(v)
                if "e" not in s and "." not in s:
                    v = s + ".0"
            cp.set(self.__config__, k, str(v))
        with open(filename, "w", encoding="utf-8") as f:
            cp.write(f)

    def load(self, filename):
        cp = configparser.ConfigParser()
        cp.read(filename)
        if not cp.has_section(self.__config__):
            sys.stderr.write('ERROR: File "' + filename + '" is not a valid configuration file for the '
                             'selected task: Missing section [' + self.__config__ + ']!\n')
            sys.exit(1)
        for k, v in cp.items(self.__config__):
            self.__dict__[k] = self._auto_cast(v)


class TaggerConfig(Config):
    """reference: cube/io_utils/config.py:104-124"""

    def __init__(self, filename=None, verbose=False):
        super().__init__()
        self.char_emb_size = 256
        self.char_filter_size = 512
        self.char_layers = 3
        self.word_emb_size = 256
        self.lang_emb_size = 64
        self.cnn_filter = 512
        self.cnn_layers = 5
        self.external_proj_size = 300
        self.lm_model = 'xlm-roberta-base'
        self._valid = True
        if filename is not None:
            self.load(filename)


class ParserConfig(Config):
    """reference: cube/io_utils/config.py:127-156"""

    def __init__(self, filename=None, verbose=False):
        super().__init__()
        self.char_emb_size = 256
        self.char_filter_size = 512
        self.char_layers = 5
        self.word_emb_size = 256
        self.lang_emb_size = 64
        self.cnn_filter = 512
        self.cnn_layers = 5
        self.aux_softmax_location = 5
        self.pre_parser_size = 500
        self.head_size = 100
        self.label_size = 200
        self.lm_model = 'xlm-roberta-base'
        self.external_proj_size = 2
        self.rhl_win_size = 2
        self.rnn_size = 200
        self.rnn_layers = 3
        self._valid = True
        if filename is not None:
            self.load(filename)


class LemmatizerConfig(Config):
    """reference: cube/io_utils/config.py:159-178"""

    def __init__(self, filename=None, verbose=False):
        super().__init__()
        self.encoder_layers = 2
        self.encoder_size = 200
        self.decoder_layers = 2
        self.decoder_size = 400
        self.att_proj_size = 100
        self.upos_emb_size = 100
        self.lang_emb_size = 100
        self.char_emb_size = 100
        self._valid = True
        if filename is not None:
            self.load(filename)


class TokenizerConfig(Config):
    """reference: cube/io_utils/config.py:81-101"""

    def __init__(self, filename=None, verbose=False):
        super().__init__()
        self.cnn_filter = 512
        self.lang_emb_size = 100
        self.cnn_layers = 5
        self.rnn_size = 50
        self.rnn_layers = 2
        self.external_proj_size = 2
        self.no_space_lang = False
        if filename is not None:
            self.load(filename)
        self._valid = True

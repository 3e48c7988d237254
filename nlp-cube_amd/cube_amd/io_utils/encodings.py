"""Vocabulary maps that size and index every table on the parser path.

Host-side mirror of the reference `Encodings` (cube/io_utils/encodings.py:8-253):
same attribute names, same on-disk text format (`LANGS / LABELS / CHARACTERS /
WORDS / UPOS / XPOS / ATTRS` sections of `key<TAB>id` rows), so a released
`*.encodings` file loads unchanged.  `compute` (training-time vocabulary
building, encodings.py:26-140) is out of scope for the inference path; a
minimal equivalent is kept so synthetic models can be written to disk.
"""


class Encodings:
    _SECTIONS = ("label2int", "char2int", "word2int", "upos2int", "xpos2int", "attrs2int")

    def __init__(self, verbose=False):
        self.word_list = {}
        self.hol_word_list = []
        self.char2int = {}
        self.label2int = {}
        self.labels = []
        self.word2int = {}
        self.upos2int = {}
        self.xpos2int = {}
        self.attrs2int = {}
        self.upos_list = []
        self.xpos_list = []
        self.attrs_list = []
        self.characters = []
        self.verbose = verbose
        self.num_langs = 0

    @staticmethod
    def _read_section(f, split_header_on_space_index=1):
        header = f.readline()
        n = int(header.split(" ")[split_header_on_space_index])
        table = {}
        inverse = [""] * n
        for _ in range(n):
            parts = f.readline().split("\t")
            key, value = parts[0], int(parts[1])
            table[key] = value
            if 0 <= value < n:
                inverse[value] = key
        return table, inverse

    def load(self, filename):
        """reference: cube/io_utils/encodings.py:149-229."""
        with open(filename, "r", encoding="utf8") as f:
            self.num_langs = int(f.readline().strip().split(" ")[-1])
            self.label2int, self.labels = self._read_section(f)
            self.char2int, self.characters = self._read_section(f)
            self.word2int, _ = self._read_section(f)
            self.upos2int, self.upos_list = self._read_section(f)
            self.xpos2int, self.xpos_list = self._read_section(f)
            self.attrs2int, self.attrs_list = self._read_section(f)

    def save(self, filename):
        """reference: cube/io_utils/encodings.py:231-253."""
        with open(filename, "w", encoding="utf8") as f:
            f.write("LANGS " + str(self.num_langs) + "\n")
            for title, table in (("LABELS", self.label2int), ("CHARACTERS", self.char2int),
                                 ("WORDS", self.word2int), ("UPOS", self.upos2int),
                                 ("XPOS", self.xpos2int), ("ATTRS", self.attrs2int)):
                f.write(title + " " + str(len(table)) + "\n")
                for key, value in table.items():
                    f.write(str(key) + "\t" + str(value) + "\n")

    # -- construction from explicit symbol lists (synthetic models / tests) ------------
    @classmethod
    def from_lists(cls, chars, words, upos, xpos, attrs, labels, num_langs):
        """Build tables with the reserved entries the reference's `compute` creates
        (encodings.py:30-56): id 0 '<PAD>', id 1 '<UNK>' everywhere and id 2 ' '
        for characters.  (The reference's in-memory `attrs_list[1]` is '<PAD>', :47-48; a
        saved-and-reloaded table — what inference uses, :217-228 — has '<UNK>', as here.)"""
        e = cls()
        e.num_langs = num_langs

        def fill(table, inverse, symbols, reserved):
            for s in reserved + [x for x in symbols if x not in reserved]:
                if s not in table:
                    table[s] = len(table)
                    if inverse is not None:
                        inverse.append(s)

        fill(e.char2int, e.characters, list(chars), ["<PAD>", "<UNK>", " "])
        fill(e.word2int, e.hol_word_list, list(words), ["<PAD>", "<UNK>"])
        fill(e.upos2int, e.upos_list, list(upos), ["<PAD>", "<UNK>"])
        fill(e.xpos2int, e.xpos_list, list(xpos), ["<PAD>", "<UNK>"])
        fill(e.attrs2int, e.attrs_list, list(attrs), ["<PAD>", "<UNK>"])
        fill(e.label2int, e.labels, list(labels), ["<PAD>", "<UNK>"])
        return e

"""Stand-in for gensim.corpora.dictionary.Dictionary: a dict {id: str(id)}."""


class Dictionary(dict):
    @classmethod
    def from_corpus(cls, corpus, id2word=None):
        max_id = -1
        for doc in corpus:
            for wid, _cnt in doc:
                if wid > max_id:
                    max_id = int(wid)
        return cls({i: str(i) for i in range(max_id + 1)})

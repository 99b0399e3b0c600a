"""Differential MLSA filtering (kwiiyatta.filter of the reference,
/root/reference/kwiiyatta/filter/mlsa.py:9-30) -- SURVEY.md section 8(f)-2,
outside this round's hot path (the north star names WORLD overlap-add
resynthesis).  The entry point exists so that the package surface matches."""


def apply_mlsa_filter(wav, mcep):
    raise NotImplementedError(
        'apply_mlsa_filter (diffVC / MLSA path) is not implemented in kwiiyatta_amd yet; '
        'use the WORLD synthesis path (convert(..., diffvc=False) / --no-diffvc)')


__all__ = ['apply_mlsa_filter']

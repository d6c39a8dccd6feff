"""Writes tests/golden/esper_inchi_all.tsv: the identity columns (isomeric SMILES, canonical SMILES, InChI, molar weight)
of ALL rows of the reference's raw Esper et al. 2023 table, gnnepcsaft/data/esper2023/raw/SI_pcp-saft_parameters.csv
(data, not code).  The InChI formula / connectivity / hydrogen layers and the molar weight are reference-held facts
about every molecule, independent of RDKit: tests/test_featurize_inchi_cpu.py pins the SMILES featuriser against them.

Run (in the build container, where /root/reference exists):  python tests/golden/make_esper_inchi_fixture.py
"""
import csv
import os

SRC = "/root/reference/gnnepcsaft/data/esper2023/raw/SI_pcp-saft_parameters.csv"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "esper_inchi_all.tsv")


def main():
    rows = list(csv.DictReader(open(SRC), delimiter="\t"))
    with open(DST, "w") as fh:
        fh.write("# %d rows of the reference's raw Esper et al. 2023 table (gnnepcsaft/data/esper2023/raw/"
                 "SI_pcp-saft_parameters.csv): columns isomeric_smiles, canonical_smiles, inchi, molarweight (data, not code)\n"
                 % len(rows))
        for r in rows:
            fh.write("\t".join([r["isomeric_smiles"], r["canonical_smiles"], r["inchi"], r["molarweight"]]) + "\n")
    print("wrote", DST, len(rows))


if __name__ == "__main__":
    main()

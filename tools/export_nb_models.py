"""Writes the naive-Bayes models of the stubborn rule as JSON for SemanticMapper(nb_models=...).

    python tools/export_nb_models.py <models.pkl> <out.json>

The reference stores {goal id: sklearn GaussianNB} as a pickle
(sem_objnav/obj_nav/nb_models/<model_name>.pkl, written by train_stubborn_nb_models.py).
Unpickling executes code from the file: run this only on a pickle you trust, in an environment
with the scikit-learn version that wrote it.  Only the four fitted arrays are kept.
"""
import json
import pickle
import sys


def main(src, dst):
    with open(src, "rb") as fh:
        models = pickle.load(fh)
    out = {}
    for goal, m in models.items():
        out[str(int(goal))] = {"classes_": [int(c) for c in m.classes_],
                               "class_prior_": [float(p) for p in m.class_prior_],
                               "theta_": [[float(x) for x in row] for row in m.theta_],
                               "var_": [[float(x) for x in row] for row in m.var_]}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

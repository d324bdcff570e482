"""Drop-in import surface for the reference's (un-vendored) `utils` package -- only what the ConceptHash
encode-and-retrieve path touches: utils.hashing, utils.io, utils.misc, utils.metrics, utils.datasets, utils.transforms."""

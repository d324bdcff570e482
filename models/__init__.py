"""Drop-in import surface for the reference's `models.*` dotted names (only what the ConceptHash path needs)."""

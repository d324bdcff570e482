"""Drop-in import surface for the reference's `trainers.*` dotted names (ConceptHash evaluation path only)."""

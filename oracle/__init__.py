"""Test infrastructure: CPU oracle for the VectorBase kNN path.  Never imported by the product package."""

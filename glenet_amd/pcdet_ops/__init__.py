"""Mirror of the reference's `pcdet.ops` operator packages."""

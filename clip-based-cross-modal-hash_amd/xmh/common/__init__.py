"""Mirror of the reference's ``common`` package: ``common.register`` and ``common.calc_utils``."""

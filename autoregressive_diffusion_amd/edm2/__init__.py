"""Drop-in namespace with the reference's `edm2.*` import surface, MI355X-native underneath."""
